#!/usr/bin/env python3
"""
bench.py -- objective evaluations per second of the MI355X swarm generation.

    python bench.py --gpus N --steps K --warmup W          (N > 1: launches its own N ranks)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W

Either form runs one process per GPU.  With WORLD_SIZE unset and --gpus N > 1 this script is the
launcher: before touching any GPU it starts N rank processes of itself (RANK / LOCAL_RANK /
WORLD_SIZE / MASTER_ADDR / MASTER_PORT set), relays rank 0's JSON line and exits non-zero if any
rank fails or the whole run outlives --launch-timeout (it then names the ranks still alive and
ends exactly the processes it started).  No PyTorch in any of it: the ranks find each other over
standard-library sockets (nmrfit_amd/rendezvous.py) and exchange through RCCL inside
libnmrfit_amd.so.

N > 1 is an RCCL measurement or it is an error: if the exchange that ran was not RCCL (library
missing, communicator creation failing) -- and NMRFIT_BENCH_BACKEND=host was not asked for
explicitly -- the line is still printed, measured over the host-staged exchange, but carries an
"error" field and the exit code is 5.  Every rank also carries its own deadline (the driver
launches the ranks under torch.distributed.run, not under this script's launcher): a rank still
inside the rendezvous / ncclCommInitRank / the first collective after --launch-timeout seconds
says where it is stuck on stderr and exits 124, which makes the launcher end the others.

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
roofline: from the objective kernel's own durations INSIDE the timed loop (kernel <= step is
          asserted).  `roofline` is the binding resource (fp64 vector-ALU issue: issued wave64 VALU instructions
          per second against 1024 SIMDs x 2.4 GHz / 4 cycles), `roofline_streaming` the metric's "HBM GB/s vs peak"
          under the streaming-operand model (an effective, L2-served rate).
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

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
HBM_COPY_GBS = 6290.0          # ... and what a float4 copy measures on it (79 %)
SIMDS = 1024                   # 256 CUs x 4 SIMDs
PEAK_CLOCK_MHZ = 2400.0        # MI355X_MICROARCH.md
LAUNCHER_GRACE_S = 15.0        # self-launcher: how much later than the ranks' own watchdogs its deadline falls
FP64_ISSUE_CYCLES = 4.0        # one wave64 fp64 VALU instruction occupies a SIMD's issue port for 4 cycles
VALU_PER_UNIT_C3 = 5.76        # fp64 VALU instructions per (particle, point, peak) of the headline kernel at C3 (profiles/r05/bench_c3_pmc_summary.json)
PMC_SUMMARIES = [os.path.join("profiles", r, "bench_c3_pmc_summary.json") for r in ("r06", "r05", "r04", "r03", "r02", "r01")]
FARFIELD_PMC_SUMMARIES = [os.path.join("profiles", r, "farfield_c3_pmc_summary.json") for r in ("r06", "r05", "r04", "r03")]


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
    ap.add_argument("--no-other-configs", dest="other_configs", action="store_false",
                    help="do not time the C2 and C5 shapes (BASELINE configs 2 and 5; kernel only, after the timed "
                         "region).  tools/profile.sh passes this so that every objective_kernel<0,false,0> launch of a "
                         "profiled run has the C3 shape and the rocprofv3 --stats average of that kernel is the number "
                         "in roofline.kernel_ms")
    ap.add_argument("--no-pmc", dest="pmc", action="store_false",
                    help="do not run the rocprofv3 --pmc child passes (N = 1, workload C3: three short passes of this "
                         "same script under `rocprofv3 --kernel-trace --pmc ...`, started BEFORE this process touches "
                         "the GPU, give the line its physical HBM traffic and VALU counters live; without them -- or if "
                         "rocprofv3 is missing -- those fields come from the committed passes under profiles/ and say so)")
    ap.add_argument("--kernel-only", choices=["sparse", "dense"], default=None,
                    help="child mode (used by this script itself for the `variants` entry and the PMC passes of the "
                         "every-unit and dense-spectrum cases): no swarm, no CPU legs -- time the objective kernel alone "
                         "on the workload's synthetic swarm (sparse: SURVEY 8(d) positions; dense: broad overlapping "
                         "lines) for every variant of --variants and print one small JSON line")
    ap.add_argument("--variants", default="0", help="--kernel-only: comma-separated NMRFIT_VARIANT_* numbers")
    ap.add_argument("--launch-timeout", type=float, default=300.0, metavar="SECONDS",
                    help="N > 1: deadline of the self-launcher for the whole run, and of every rank for reaching the "
                         "timed region (rendezvous, RCCL communicator, first collective); 0 disables")
    return ap.parse_args()


# ---- launcher (N > 1 without a launcher's environment) -------------------------------------------
def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def self_launch(n, launch_timeout):
    """Start n rank processes of this script, one per GPU.  The parent never touches the GPU (it
    loads neither the library nor HIP): it only waits, relays rank 0's output, and makes sure a
    failed rank -- or a run that outlives `launch_timeout` seconds -- takes the others down instead
    of leaving them blocked in a collective.  It never replaces itself or any rank with another
    program: it ends the child processes it started, by pid, and exits."""
    port = _free_port()
    token = "bench%d_%d" % (os.getpid(), int(time.time() * 1e3) & 0xFFFFFFF)
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), NMRFIT_RDZV_TOKEN=token)
        # (HSA_ENABLE_IPC_MODE_LEGACY=0 -- dmabuf IPC, the only kind this pool's driver offers -- is set by
        # nmrfit_amd/_cabi.py before the library loads, for every launcher alike)
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL, stderr=None))
    rc = 0
    out0 = b""
    t_start = time.monotonic()
    try:
        pending = set(range(n))
        while pending:
            # the backstop: every rank carries the same deadline for reaching the timed region and says where it
            # was stuck (more useful than anything the launcher can say), so the launcher waits a little longer
            if launch_timeout > 0 and time.monotonic() - t_start > launch_timeout + LAUNCHER_GRACE_S:
                alive = [r for r in sorted(pending) if procs[r].poll() is None]
                sys.stderr.write("bench.py: launch timeout: %d rank(s) still running after %.0f s: %s -- ending them\n"
                                 % (len(alive), launch_timeout + LAUNCHER_GRACE_S, alive))
                rc = 124
                break
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


class PowerProbe:
    """Socket power and shader clock of ONE device from its hwmon files (plain reads of sysfs, no child process):
    /sys/bus/pci/devices/<pci id>/hwmon/hwmon*/{power1_input (or power1_average), power1_cap, freq1_input}.  Sampled by
    the main thread between batches of the pre-heat launches: under this fp64 load an MI355X sits at its power cap
    and the clock is what is left -- the line should say so with this run's own numbers."""

    def __init__(self, pci):
        import glob
        self.dir, self.samples = None, []
        try:
            cand = glob.glob("/sys/bus/pci/devices/%s/hwmon/hwmon*" % pci.strip().lower())
            self.dir = cand[0] if cand else None
        except Exception:
            self.dir = None

    def _read(self, *names):
        for n in names:
            try:
                with open(os.path.join(self.dir, n)) as f:
                    return float(f.read().strip())
            except Exception:
                continue
        return None

    def sample(self):
        if self.dir is None:
            return
        w, hz = self._read("power1_input", "power1_average"), self._read("freq1_input")
        if w is not None and hz is not None:
            self.samples.append((w * 1e-6, hz * 1e-6))

    def summary(self):
        if self.dir is None or not self.samples:
            return None
        cap = self._read("power1_cap")
        tail = self.samples[len(self.samples) // 2:]          # the second half of the pre-heat: the loaded state
        return {"socket_power_w": sum(q[0] for q in tail) / len(tail), "socket_power_w_max": max(q[0] for q in self.samples),
                "power_cap_w": None if cap is None else cap * 1e-6,
                "sclk_mhz": sum(q[1] for q in tail) / len(tail), "sclk_mhz_nominal": 2400.0, "samples": len(self.samples),
                "source": self.dir,
                "note": "hwmon readings between batches of the pre-heat launches (the same kernel on the same positions, "
                        "right before the timed region; the power reading is a running average that is still rising after "
                        "half a second -- it settles at 1320-1380 W, profiles/r04/power_and_clock_under_load.txt): the part "
                        "is power-limited under this fp64 load and the clock is what is left, which is why the kernel time "
                        "differs from box to box"}


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
    # BASELINE config 1 (the reference's own CPU-runnable case): 50 particles x 4096 points x 6 peaks, the numpy
    # oracle one call per particle on one core -- exactly what pyswarm makes the reference do per generation
    try:
        c1 = synth.CONFIGS["C1"]
        sp1 = synth.make_spectrum(c1.N, c1.P, seed=1)
        X1 = synth.make_swarm(sp1["lower"], sp1["upper"], c1.S, seed=2, x_true=sp1["x_true"])
        t1 = time.perf_counter()
        gens1 = 0
        while gens1 < 20 and time.perf_counter() - t1 < 2.0:
            for i in range(c1.S):
                onp.objective(X1[i], sp1["w"], sp1["u"], sp1["v"], sp1["weights"])
            gens1 += 1
        dt1 = time.perf_counter() - t1
        out["c1_numpy"] = {"value": gens1 * c1.S * c1.N * c1.P / dt1, "cores": 1, "ms_per_generation": dt1 / gens1 * 1e3,
                           "sample": "C1: %d generations of %d particles x %d points x %d peaks, numpy oracle, %.2f s"
                                     % (gens1, c1.S, c1.N, c1.P, dt1)}
    except Exception as e:
        out["c1_numpy"] = {"error": repr(e)}
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


# ---- live PMC passes (before any GPU call of this process) ----------------------------------------
GEN_LOOP = ["--steps", "5", "--warmup", "2", "--cpu-seconds", "0", "--no-extras", "--no-other-configs", "--no-pmc",
            "--preheat-seconds", "0.2"]
SQ = ["SQ_BUSY_CYCLES", "SQ_INSTS_VALU", "SQ_ACTIVE_INST_VALU", "SQ_WAVES"]
PMC_PASSES = (
    # (name, bench args of the child, counters, needs the A/B library) -- one rocprofv3 run each: counters of different
    # blocks are collected in their own passes, with --kernel-trace only (MI355X_MICROARCH.md, rocprofv3 PMC slots)
    ("sq", GEN_LOOP, ["SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "SQ_INSTS_VALU", "SQ_ACTIVE_INST_VALU", "SQ_WAIT_INST_ANY",
                      "SQ_ACTIVE_INST_ANY", "SQ_WAVES"], False),
    ("fetch", GEN_LOOP, ["FETCH_SIZE", "GRBM_GUI_ACTIVE"], False),
    ("write", GEN_LOOP, ["WRITE_SIZE", "TCC_HIT_sum", "TCC_MISS_sum"], False),
    ("sq_farfield", GEN_LOOP + ["--variant", "6"], SQ, False),
    # round 5: the cases where every unit is evaluated -- NOSKIP (an A/B kernel: libnmrfit_amd_ab.so) on the sparse
    # spectrum, the headline and far-field kernels on the dense one -- objective launches only
    ("sq_noskip", ["--kernel-only", "sparse", "--variants", "2"], SQ, True),
    ("sq_dense", ["--kernel-only", "dense", "--variants", "0"], SQ, False),
    ("sq_dense_farfield", ["--kernel-only", "dense", "--variants", "6"], SQ, False),
)


def ab_library():
    """Path of the A/B library (the product + BASELINE / NOSKIP / SINGLE / QUAD / STAGED kernels), or None."""
    p = os.path.join(ROOT, "nmrfit_amd", "lib", "libnmrfit_amd_ab.so")
    return p if os.path.exists(p) else None


def live_pmc_passes(timeout_s=90.0):
    """Run this script's C3 generation loop in child processes under `rocprofv3 --kernel-trace --pmc ...`
    (the program itself after `--`; nothing else in between) and return the objective kernel's mean
    counters per pass, with the child's own in-run kernel time and clock.  The parent has not touched
    the GPU yet (it starts children only, never replaces itself).  Any failure returns what was
    collected plus an `errors` list: the caller then falls back to the committed passes."""
    import csv
    import glob
    import shutil
    import tempfile
    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    out = {"errors": []}
    if not os.path.exists(exe):
        out["errors"].append("rocprofv3 not found")
        return out
    for name, child_args, counters, needs_ab in PMC_PASSES:
        env = dict(os.environ, TMPDIR="/tmp")
        if needs_ab:
            if ab_library() is None:
                out["errors"].append("%s: libnmrfit_amd_ab.so not built" % name)
                continue
            env["NMRFIT_LIB"] = ab_library()
        try:
            tmp = tempfile.mkdtemp(prefix="nmrfit_pmc_", dir="/tmp")
        except OSError as e:
            out["errors"].append("%s: %r" % (name, e))
            break
        cmd = [exe, "--kernel-trace", "--pmc"] + counters + ["--output-format", "csv", "-d", tmp, "--",
               sys.executable, os.path.abspath(__file__)] + child_args
        try:
            # its own process group: on a time-out the profiler AND the program under it are ended (by the
            # exact group id started here), not just the profiler's front end
            pr = subprocess.Popen(cmd, cwd=tmp, env=env, stdout=subprocess.PIPE,
                                  stderr=subprocess.PIPE, text=True, start_new_session=True)
            try:
                so, se = pr.communicate(timeout=timeout_s)
            except subprocess.TimeoutExpired:
                import signal
                os.killpg(pr.pid, signal.SIGTERM)
                try:
                    pr.communicate(timeout=10)
                except subprocess.TimeoutExpired:
                    os.killpg(pr.pid, signal.SIGKILL)
                    pr.communicate()
                out["errors"].append("%s: timed out after %.0f s" % (name, timeout_s))
                break      # something is wrong with the profiler here: do not try the remaining passes
            lines = [l for l in so.splitlines() if l.startswith("{")]
            if pr.returncode != 0 or not lines:
                out["errors"].append("%s: rc %d %s" % (name, pr.returncode, se[-300:]))
                continue
            child = json.loads(lines[-1])
            fs = glob.glob(os.path.join(tmp, "**", "*_counter_collection.csv"), recursive=True)
            if not fs:
                out["errors"].append("%s: no counter file" % name)
                continue
            agg = {}
            with open(fs[0]) as fh:
                for row in csv.DictReader(fh):
                    if "objective_kernel" in row["Kernel_Name"]:
                        agg.setdefault(row["Counter_Name"], []).append(float(row["Counter_Value"]))
            ko = child.get("kernel_only")
            kms = (list(ko["variants"].values())[0]["kernel_ms"] if ko else
                   child["kernel_ms"]["mean"] if child.get("kernel_ms") else None)
            out[name] = {"counters": {k: sum(v) / len(v) for k, v in agg.items()},
                         "launches": max((len(v) for v in agg.values()), default=0),
                         "kernel_ms_in_child": kms,
                         "clock_mhz_in_child": child.get("roofline", {}).get("clock_mhz_in_run")}
        except Exception as e:     # a profiler hiccup must not cost the run its headline
            out["errors"].append("%s: %r" % (name, e))
        finally:
            shutil.rmtree(tmp, ignore_errors=True)
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


VARIANT_NAMES = {0: "default", 1: "baseline", 2: "noskip", 3: "single", 4: "quad", 5: "staged", 6: "farfield", 7: "norec"}


def kernel_only(args):
    """Child mode: the objective kernel alone on the workload's synthetic swarm, for each variant asked for."""
    json_fd = os.dup(1)
    os.dup2(2, 1)
    from nmrfit_amd import _cabi, synth
    from nmrfit_amd.equations import Evaluator
    cfg = synth.CONFIGS[args.workload]
    S, N, P = (args.swarm_per_gpu or cfg.S), cfg.N, cfg.P
    spec = synth.make_spectrum(N, P, seed=1)
    if args.kernel_only == "dense":
        X = synth.make_dense_swarm(S, P, seed=5, w_lo=float(spec["w"].min()), w_hi=float(spec["w"].max()))
    else:
        X = synth.make_swarm(spec["lower"], spec["upper"], S, seed=2, x_true=spec["x_true"])
    out = {"spectrum": args.kernel_only, "library": os.path.basename(_cabi.lib_path()), "variants": {}}
    with Evaluator(spec["w"], spec["u"], spec["v"], spec["weights"], device=0) as ev:
        d_x, d_f = ev.dev_alloc(X.nbytes), ev.dev_alloc(S * 8)
        ev.upload(d_x, X)
        f0 = None
        for v in [int(x) for x in args.variants.split(",")]:
            ev.set_variant(v)
            ms = time_objective(ev, S, P, d_x, d_f, max(5, min(args.steps, 20)))
            f = ev.download(d_f, (S,))
            f0 = f if f0 is None else f0
            out["variants"][VARIANT_NAMES[v]] = {
                "kernel_ms": ms, "units_per_s": float(S) * N * P / (ms * 1e-3),
                "max_rel_diff_vs_first": float(np.max(np.abs(f - f0) / np.maximum(np.abs(f0), 1e-6)))}
        ev.dev_free(d_x)
        ev.dev_free(d_f)
    os.write(json_fd, (json.dumps({"kernel_only": out}) + "\n").encode())


def kernel_only_child(spectrum, variants, use_ab, timeout_s=120.0, extra=()):
    """Run `bench.py --kernel-only` in a child process (with the A/B library when asked) and return its JSON."""
    env = dict(os.environ)
    if use_ab:
        if ab_library() is None:
            return {"error": "libnmrfit_amd_ab.so not built (nmrfit_amd/csrc/build.sh --ab)"}
        env["NMRFIT_LIB"] = ab_library()
    cmd = [sys.executable, os.path.abspath(__file__), "--kernel-only", spectrum, "--variants",
           ",".join(str(v) for v in variants)] + list(extra)
    try:
        pr = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=timeout_s)
    except subprocess.TimeoutExpired:
        return {"error": "timed out after %.0f s" % timeout_s}
    lines = [l for l in pr.stdout.splitlines() if l.startswith("{")]
    if pr.returncode != 0 or not lines:
        return {"error": "rc %d %s" % (pr.returncode, pr.stderr[-300:])}
    return json.loads(lines[-1])["kernel_only"]


def main():
    args = parse()
    if args.kernel_only:
        return kernel_only(args)
    world_env = os.environ.get("WORLD_SIZE")
    if world_env is None and args.gpus > 1:
        raise SystemExit(self_launch(args.gpus, args.launch_timeout))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(world_env or "1")
    if args.gpus != world:
        raise SystemExit("--gpus %d does not match WORLD_SIZE=%d" % (args.gpus, world))
    # stdout carries ONE JSON line and nothing else.  Libraries print there too -- RCCL writes a version banner
    # (and, with NCCL_DEBUG=WARN, its warnings) to stdout at communicator creation, fit() prints pyswarm's
    # "Stopping search" line -- so the original stdout is kept aside for the line, and file descriptor 1 (what C
    # libraries and Python's print use from here on) is pointed at stderr.
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)

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

    # ---- live PMC passes: children under rocprofv3, still before this process touches the GPU ----
    pmc_live = None
    if (rank == 0 and world == 1 and args.pmc and args.workload == "C3" and args.variant == 0
            and args.swarm_per_gpu in (0, cfg.S) and not under_profiler()):
        try:
            pmc_live = live_pmc_passes()
        except Exception as e:      # the counters are an extra: never at the price of the headline
            pmc_live = {"errors": ["live_pmc_passes: %r" % (e,)]}

    from nmrfit_amd import _cabi, pso
    from nmrfit_amd.equations import Evaluator

    # N > 1: one rank per GPU, RCCL through the C-ABI.  Rehearsal knobs (not used by the
    # driver): NMRFIT_BENCH_FORCE_DIST=1 takes the RCCL path with a single rank;
    # NMRFIT_BENCH_BACKEND=host (alias: gloo) runs several ranks on ONE GPU with the record staged
    # through the host over sockets (RCCL refuses two ranks on one device).
    backend = os.environ.get("NMRFIT_BENCH_BACKEND", "rccl").lower()
    backend = {"nccl": "rccl", "gloo": "host"}.get(backend, backend)
    use_dist = world > 1 or os.environ.get("NMRFIT_BENCH_FORCE_DIST") == "1"
    from nmrfit_amd import rendezvous
    state = {"first": True}
    n_visible = _cabi.device_count()
    visible_env = rendezvous.visible_devices_env()
    place = {"device": local_rank, "pci": "-", "note": ""}      # filled in below, read by the watchdog's message

    def where():
        return "(HIP device %d of %d visible, PCI %s, %s, world %d)" % (
            place["device"], n_visible, place["pci"], visible_env or "no *_VISIBLE_DEVICES set", world)
    # Every rank's own deadline for reaching the timed region (N > 1): the driver launches the ranks
    # under torch.distributed.run, so this script's launcher is not there to notice a rank that never
    # comes out of the rendezvous / ncclCommInitRank / the first collective.
    # test hook (tests/test_rendezvous_cpu.py): NMRFIT_BENCH_TEST_STALL="pre:<rank|all>:<seconds>" makes a rank
    # sleep before its watchdog exists (only the launcher's deadline can end it), "in:..." inside it
    stall = os.environ.get("NMRFIT_BENCH_TEST_STALL", "").split(":")
    stall_s = float(stall[2]) if len(stall) == 3 and stall[1] in ("all", str(rank)) else 0.0
    if stall_s and stall[0] == "pre":
        time.sleep(stall_s)
    dog = rendezvous.Watchdog(args.launch_timeout if use_dist else 0, "start-up", rank=rank, describe=where)
    dog.__enter__()
    if stall_s and stall[0] == "in":
        dog.phase = "test stall"
        time.sleep(stall_s)

    dog.phase = "device selection"
    if use_dist and (backend == "host" or os.environ.get("NMRFIT_BENCH_SHARE_GPU") == "1"):
        device, device_note = local_rank % max(1, n_visible), "rehearsal: several ranks share one card"
    else:
        # LOCAL_RANK, or device 0 when the launcher shows this rank one device only (HIP_VISIBLE_DEVICES /
        # ROCR_VISIBLE_DEVICES isolation); any other mismatch ends this rank here, before the rendezvous, with
        # a message that names the variables -- the launcher then ends the others
        try:
            device, device_note = rendezvous.pick_device(n_visible, local_rank)
        except RuntimeError as e:
            sys.stderr.write("bench.py rank %d/%d: %s\n" % (rank, world, e))
            raise SystemExit(6)
    if device_note:
        sys.stderr.write("bench.py rank %d/%d: %s\n" % (rank, world, device_note))
    try:      # looked up now: the watchdog thread must not make HIP calls while the main thread is stuck in one
        pci = _cabi.device_pci_bus_id(device)
    except _cabi.NmrfitError as e:
        pci = "unknown (%s)" % e
    place.update(device=device, pci=pci, note=device_note)

    dog.phase = "context creation"
    ev = Evaluator(spec["w"], spec["u"], spec["v"], spec["weights"], device=device)
    ev.set_variant(args.variant)
    sw = pso.DeviceSwarm(ev, spec["lower"], spec["upper"], swarmsize=S_local * world, offset=rank * S_local,
                         S_local=S_local, seed=1234, minstep=-1.0, minfunc=-1.0)   # never stop while timing
    ex = None
    exchange_desc = "none"
    rccl_failure = None
    rccl_info = None
    channel = None
    host_requested = backend == "host"
    if use_dist:
        dog.phase = "socket rendezvous (nmrfit_amd/rendezvous.py)"
        channel = rendezvous.Channel()        # the star of sockets the ranks bootstrap over
    if use_dist and backend == "rccl":
        dog.phase = "RCCL communicator creation"
        if os.environ.get("NMRFIT_BENCH_TEST_RCCL_MISSING_ON") == str(rank):   # test hook: RCCL missing on ONE rank
            os.environ["NMRFIT_RCCL_LIB"] = "/nonexistent/librccl.so"
        sys.stderr.write("bench.py rank %d/%d: creating the RCCL communicator on HIP device %d %s\n"
                         % (rank, world, device, where()))
        sys.stderr.flush()
        try:
            # init_timeout=0: this run's own watchdog (dog) already covers the collective creation
            ex = pso.RcclExchange(ev, channel=channel, init_timeout=0, verbose=True)
        except _cabi.NmrfitError as e:
            # RCCL unavailable (library missing on some rank: every rank raises alike, see
            # RcclExchange; no unique id; communicator creation failing): say so loudly and still
            # measure, with the candidate record staged through the host -- the line then carries an
            # "error" field and the exit code is non-zero.
            rccl_failure = str(e)
            sys.stderr.write("bench.py rank %d: RCCL exchange unavailable (%s)\n" % (rank, e))
        # every rank takes the same path: RCCL only if every rank has a communicator
        dog.phase = "agreement on the exchange path"
        oks = channel.all_gather(b"\x01" if ex is not None else b"\x00")
        if not all(o == b"\x01" for o in oks):
            if ex is not None:
                ex.close()
                ex = None
            rccl_failure = rccl_failure or "RCCL failed on rank(s) %s" % [i for i, o in enumerate(oks) if o != b"\x01"]
            backend = "host"
    if use_dist and backend == "rccl":
        dog.phase = "first RCCL collective"
        info = ex.info()
        # how many ranks RCCL itself saw: an all-reduce of ones over the communicator
        seen = int(round(float(ex.all_reduce([1.0], "sum")[0])))
        # which GPU every rank ended up on (device index as this rank sees it, PCI id, its visibility variables)
        placed = [json.loads(b.decode()) for b in channel.all_gather(json.dumps(
            {"rank": rank, "local_rank": local_rank, "device": device, "visible_devices": n_visible, "pci": pci,
             "env": visible_env, "note": device_note}).encode())]
        rccl_info = {"nranks": info["world"], "ranks_counted_by_all_reduce": seen, "version": info["rccl_version"],
                     "rank0": ex.describe(), "placement": placed,
                     "distinct_pci_ids": len(set(q["pci"] for q in placed)),
                     # which library answered: anything but the system's librccl is a rehearsal, not a measurement
                     "library": os.environ.get("NMRFIT_RCCL_LIB") or "librccl (default search path)"}
        sw.set_comm(ex)                       # the all-gather now happens inside nmrfit_pso_step
        exchange_desc = "ncclAllGather of %d doubles per generation inside nmrfit_pso_step (RCCL %s)" % (
            D + 1, info["rccl_version"])

        def step():
            sw.step()
    elif use_dist:
        from tests.swarm_support import SocketExchange   # (rehearsal / RCCL-failed path: test infrastructure)
        ex = SocketExchange(channel=channel)
        exchange_desc = "host-staged all-gather of %d doubles per generation (sockets%s)" % (
            D + 1, "; RCCL FAILED: " + rccl_failure if rccl_failure else "; rehearsal")

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

    dog.phase = "swarm init and warm-up generations"
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
    power = PowerProbe(pci)
    while time.perf_counter() - t0 < args.preheat_seconds:
        for _ in range(8):
            ev.objective_batch_dev(S_local, P, d_x, d_f)
        ev.synchronize()
        heat_launches += 8
        power.sample()
    ev.prof_enable(args.steps)

    dog.phase = "barrier before the timed region"
    barrier()
    dog.__exit__(None, None, None)           # every rank is here: from now on a hang is the driver's to time
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
        pw = power.summary() or {}
        mine = [km, dt_local, clock_mhz, float(pw.get("socket_power_w") or 0.0), float(pw.get("sclk_mhz") or 0.0)]
        lo = ex.all_reduce(mine, "min")
        hi = ex.all_reduce(mine, "max")
        ranks = {"kernel_ms_mean": {"min": float(lo[0]), "max": float(hi[0])},
                 "timed_region_s": {"min": float(lo[1]), "max": float(hi[1])},
                 "clock_mhz_in_run": {"min": float(lo[2]), "max": float(hi[2])},
                 # (0: the hwmon files of that rank's device were not readable)
                 "socket_power_w_preheat": {"min": float(lo[3]), "max": float(hi[3])},
                 "sclk_mhz_preheat": {"min": float(lo[4]), "max": float(hi[4])},
                 # what a generation costs beyond its objective kernel: the candidate launch, the all-gather, the fold
                 # -- and waiting for the slowest rank in the all-gather
                 "step_minus_slowest_kernel_ms": dt / args.steps * 1e3 - float(hi[0]),
                 "note": "spread over the ranks (every rank evaluates the same amount of work; `value` uses the slowest: "
                         "the ranks meet in the all-gather of every generation, so a GPU that holds a lower clock at its "
                         "power cap sets the pace)"}
    ms_per_step = dt / args.steps * 1e3
    st = sw.status()
    geom = ev.last_launch()
    ev.upload(d_x, sw.state()["x"])

    # ---- extras on the final swarm positions (rank 0, after the timed region) ---------------
    variants = farfield = host_ms = others = None  # (default_fit below)
    extras_errors = []
    try:
        if rank == 0 and args.variant == 0 and not args.no_extras:
            f_def = None
            variants = {}
            reps = max(5, min(args.steps, 20))
            for name, vid in (("default", _cabi.VARIANT_DEFAULT), ("farfield", _cabi.VARIANT_FARFIELD),
                              ("farfield32", _cabi.VARIANT_FARFIELD32)):
                ev.set_variant(vid)
                ms = time_objective(ev, S_local, P, d_x, d_f, reps)
                f = ev.download(d_f, (S_local,))
                if f_def is None:
                    f_def = f
                variants[name + "_ms"] = ms
                variants[name + "_max_rel_diff_vs_default"] = float(np.max(np.abs(f - f_def) / np.maximum(np.abs(f_def), 1e-6)))
            ev.set_variant(args.variant)
            # the every-unit forms are A/B kernels: not in the product library.  A child process loads the A/B library
            # (libnmrfit_amd_ab.so) and times them, with DEFAULT beside them, on the workload's synthetic swarm
            # (SURVEY 8(d) positions: the swarm before it has converged)
            eu = kernel_only_child("sparse", (0, 2, 1), use_ab=True, extra=["--workload", args.workload])
            if "error" not in eu:
                for name in ("noskip", "baseline"):
                    variants[name + "_ms"] = eu["variants"][name]["kernel_ms"]
                    variants[name + "_max_rel_diff_vs_default"] = eu["variants"][name]["max_rel_diff_vs_first"]
                variants["default_ms_in_ab_child"] = eu["variants"]["default"]["kernel_ms"]
            else:
                variants["every_unit_error"] = eu["error"]
            variants["note"] = ("objective kernel alone (mean of %d HIP-event pairs after 0.25 s of the same launches): "
                                "default / farfield on the final swarm positions, this process; noskip / baseline -- which "
                                "evaluate every (particle, point, peak) unit: DEFAULT skips out-of-window Gaussians (exact to "
                                "fp64 rounding), baseline is IEEE divide + libdevice exp2 per unit -- are A/B kernels, timed by "
                                "a child process with libnmrfit_amd_ab.so on the workload's initial synthetic swarm "
                                "(default_ms_in_ab_child: DEFAULT there).  farfield is what fit() picks at this size (see "
                                "fit_default) and never the configuration `value` is measured on" % reps)
            farfield = {"kernel_ms": variants["farfield_ms"],
                        "units_per_s": float(S_local) * N * P / (variants["farfield_ms"] * 1e-3),
                        "max_rel_diff_vs_default": variants["farfield_max_rel_diff_vs_default"]}
    except Exception as e:      # an extra must never cost the run its headline line
        extras_errors.append("variants: %r" % (e,))
        variants = farfield = None
    # the imaginary channel (fit_im=True: the reference's last-peak-only term, nmrfit/equations.py:197-209; "sum": every
    # peak's Kramers-Kronig partner) on the same positions: kernel alone, both kernels fit() can select -- never `value`
    imag = None
    try:
        if rank == 0 and args.variant == 0 and not args.no_extras:
            imag = {}
            for name, vid in (("default", _cabi.VARIANT_DEFAULT), ("farfield", _cabi.VARIANT_FARFIELD)):
                ev.set_variant(vid)
                for mode, key in ((True, "fit_im_true_ms"), ("sum", "fit_im_sum_ms")):
                    ev.set_fit_im(mode)
                    imag.setdefault(name, {})[key] = time_objective(ev, S_local, P, d_x, d_f, 8, heat_s=0.1)
            ev.set_fit_im(False)
            ev.set_variant(args.variant)
            imag["note"] = ("objective kernel with the imaginary channel, mean of 8 HIP-event pairs: closed-form Kramers-Kronig "
                            "partner (Lorentzian dispersion + Dawson's integral) instead of the reference's quadrature per "
                            "point; fit() selects the far-field kernel at this size in every mode (round 6: its all-peak form "
                            "runs three waves per SIMD -- 3.03 ms in round 5)")
    except Exception as e:      # an extra must never cost the run its headline line
        extras_errors.append("imaginary_channel: %r" % (e,))
        imag = None
    finally:
        ev.set_fit_im(False)
        ev.set_variant(args.variant)
    # the same kernels on a DENSE spectrum of the same shape (broad overlapping lines: no Gaussian window misses a
    # chunk, no peak is far from any chunk): the headline's rate is a property of the sparse-line spectrum SURVEY
    # 8(d) prescribes as much as of the kernel, so the other end of the range is reported beside it -- never `value`
    dense = None
    try:
        if rank == 0 and args.variant == 0 and not args.no_extras and args.other_configs:
            # (--no-other-configs, i.e. profiled runs: every launch of the headline kernel then has the workload's
            # own spectrum, and the profiler's average duration of that kernel is roofline.kernel_ms)
            Xd = synth.make_dense_swarm(S_local, P, seed=5, w_lo=float(spec["w"].min()), w_hi=float(spec["w"].max()))
            ev.upload(d_x, Xd)
            dense = {"spectrum": "broad overlapping lines (synth.make_dense_swarm: widths 0.3-0.8 of the span), same "
                                 "shape as the workload; objective kernel alone, after the timed region"}
            reps = max(5, min(args.steps, 10))
            for name, vid in (("default", _cabi.VARIANT_DEFAULT), ("farfield", _cabi.VARIANT_FARFIELD)):
                ev.set_variant(vid)
                ms = time_objective(ev, S_local, P, d_x, d_f, reps)
                dense[name] = {"kernel_ms": ms, "units_per_s": float(S_local) * N * P / (ms * 1e-3)}
            ev.set_variant(args.variant)
            dense["kernel_ms"] = dense["default"]["kernel_ms"]          # the kernel `value` is measured on
            dense["units_per_s"] = dense["default"]["units_per_s"]
            ev.upload(d_x, sw.state()["x"])
    except Exception as e:
        extras_errors.append("dense_spectrum: %r" % (e,))
        dense = None
    try:
        if rank == 0 and world == 1 and args.workload == "C3" and args.variant == 0 and args.other_configs:
            others = {"note": "BASELINE configs 1 (C1: the reference's CPU case, here on the GPU), 2 (C2) and 5 (C5): kernel only "
                              "(HIP events around 50 launches after 5 warm-up launches), after the timed region"}
            for name in ("C1", "C2", "C5"):
                c = synth.CONFIGS[name]
                ev.synchronize()
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
                    D2 = 4 + 3 * c.P
                    # every row streams w, u, v, weights once (+ its parameters and result);
                    # C5 also writes its residual row
                    bytes2 = B * (4 * c.N * 8) + B * D2 * 8 + B * 8 + (B * c.N * 8 if name == "C5" else 0)
                    others[name] = {"shape": {"rows": B, "grid": c.N, "peaks": c.P}, "kernel_ms": ms2,
                                    "units_per_s": float(B) * c.N * c.P / (ms2 * 1e-3),
                                    "kind": "residual_batch (R rows written)" if name == "C5" else "objective_batch",
                                    # What binds a launch of a few microseconds is ONE wave's critical path (prologue + its
                                    # chunks: DESIGN.md 4.2), neither bandwidth nor issue slots.  `frac` is the share of the
                                    # chip's fp64 issue slots the launch used, ESTIMATED with the instructions per unit of the
                                    # C3 counter pass (no counter pass of its own); the streaming-operand rate of SURVEY
                                    # 8(d)(i) is an effective rate served from L2 and is given without a fraction
                                    "roofline": {"bound": "latency",
                                                 "achieved": float(B) * c.N * c.P * VALU_PER_UNIT_C3 / 64.0 / (ms2 * 1e-3),
                                                 "peak": SIMDS * PEAK_CLOCK_MHZ * 1e6 / FP64_ISSUE_CYCLES,
                                                 "unit": "fp64 wave64 VALU instructions/s (estimated: %.2f instructions per unit, "
                                                         "the C3 counter pass's figure)" % VALU_PER_UNIT_C3,
                                                 "frac": float(B) * c.N * c.P * VALU_PER_UNIT_C3 / 64.0 / (ms2 * 1e-3)
                                                         / (SIMDS * PEAK_CLOCK_MHZ * 1e6 / FP64_ISSUE_CYCLES),
                                                 "frac_kind": "estimate",
                                                 "streaming_operand_GBps_effective": bytes2 / (ms2 * 1e-3) / 1e9,
                                                 "bytes_per_launch": bytes2,
                                                 "model": "latency-bound: the launch lasts as long as one wave needs for its "
                                                          "prologue and chunks (%d waves on 1024 SIMDs)" % ev2.last_launch()["waves"]}}
                    ev2.dev_free(dX2)
                    ev2.dev_free(df2)
                    if dR2 is not None:
                        ev2.dev_free(dR2)
    except Exception as e:      # an extra must never cost the run its headline line
        extras_errors.append("other_configs: %r" % (e,))
        others = {"error": repr(e)}
    # the reference's DEFAULT workload end to end (utils.py:177-178: 204 particles, maxiter 2000) on a 6-peak,
    # 4096-point spectrum: wall time of a whole nmrfit_amd.fit() call with the stopping rule off -- every one of
    # the 2000 generations runs -- i.e. 204 x 2001 objective evaluations, which the reference makes one numpy
    # call at a time (cpu_baseline.c1_numpy: ~0.27 ms each on this host)
    default_fit = None
    if rank == 0 and world == 1 and args.workload == "C3" and args.variant == 0 and args.other_configs:
        try:
            import nmrfit_amd
            spf = synth.make_spectrum(4096, 6, seed=1)
            dataf = synth.SynthData(spf["w"], spf["u"], spf["v"], spf["peaks"])
            optsf = {"seed": 7, "minstep": -1.0, "minfunc": -1.0, "device": device}
            import contextlib
            import io
            with contextlib.redirect_stdout(io.StringIO()):     # fit() prints pyswarm's "Stopping search: ..." line;
                nmrfit_amd.fit(dataf, list(spf["lower"]), list(spf["upper"]), summary=False,   # stdout is for the JSON line only
                               options=dict(optsf, maxiter=5))
                tf = time.perf_counter()
                rf = nmrfit_amd.fit(dataf, list(spf["lower"]), list(spf["upper"]), summary=False, options=optsf)
                dtf = time.perf_counter() - tf
            default_fit = {"shape": {"swarm": 204, "grid": 4096, "peaks": 6, "generations": 2000},
                           "wall_ms": dtf * 1e3, "us_per_generation": dtf / 2000 * 1e6,
                           "units_per_s": 204.0 * 4096 * 6 * 2001 / dtf, "error": float(rf.error),
                           "note": "one whole nmrfit_amd.fit() call with the reference's defaults (weights, context, "
                                   "device swarm, 2000 generations with the stopping tests off); small swarms are bound "
                                   "by the critical path of one wave per generation, not by throughput (DESIGN.md 4.2)"}
        except Exception as e:      # reported, never fatal for the headline
            default_fit = {"error": repr(e)}
    # ... and the same default fits DEVICE-BATCHED (round 5; nmrfit_batch_*, csrc/batch.hip): K spectra, one launch per
    # generation for all K swarms, every fit bit-identical to the lone one above
    batched_fit = None
    if rank == 0 and world == 1 and args.workload == "C3" and args.variant == 0 and args.other_configs:
        try:
            from nmrfit_amd.batch import FitBatch
            Kb = 40
            specs = [synth.make_spectrum(4096, 6, seed=100 + k % 8) for k in range(Kb)]
            spectra = [(q["w"], q["u"], q["v"], q["weights"]) for q in specs]
            batched_fit = {"shape": {"fits": Kb, "swarm": 204, "grid": 4096, "peaks": 6, "generations": 2000}}
            with FitBatch(spectra[:2], [q["lower"] for q in specs[:2]], [q["upper"] for q in specs[:2]], swarmsize=204,
                          seeds=[1, 2], device=device) as fb:      # (first use of the batch kernels: their code objects load now)
                fb.run(3, 3)
            for key, rule in (("stopping_rule_off", dict(minstep=-1.0, minfunc=-1.0)), ("stopping_rule_on", {})):
                tb = time.perf_counter()
                with FitBatch(spectra, [q["lower"] for q in specs], [q["upper"] for q in specs], swarmsize=204,
                              seeds=list(range(7, 7 + Kb)), device=device, **rule) as fb:
                    tc = time.perf_counter()
                    fb.run(2000, 64)
                    tr = time.perf_counter()
                    stb = fb.status()
                    bestb = fb.best()
                    gm = fb.geometry()
                dtb = time.perf_counter() - tb
                gens = [q["iteration"] for q in stb]
                batched_fit[key] = {"wall_ms": dtb * 1e3, "fits_per_s": Kb / dtb, "create_ms": (tc - tb) * 1e3,
                                    "run_ms": (tr - tc) * 1e3,
                                    "generations": {"min": min(gens), "max": max(gens), "mean": float(np.mean(gens))},
                                    "us_per_fit_generation": dtb * 1e6 / max(1.0, float(np.sum(gens))),
                                    "geometry": gm, "error_fit0": bestb[0][1]}
            # ... and spectra of DIFFERENT lengths in one batch (round 6; every dataset is cropped to its own region,
            # nmrfit/containers.py:112-130): lengths drawn from 3000 ... 6000 against equal lengths of the same mean
            lens = [int(n) for n in np.random.default_rng(11).integers(3000, 6001, Kb)]
            mean_len = int(round(float(np.mean(lens)) / 512.0)) * 512
            rag = {"lengths": {"min": min(lens), "max": max(lens), "mean": float(np.mean(lens))}, "equal_length": mean_len}
            for name, lengths in (("ragged", lens), ("equal", [mean_len] * Kb)):
                sps = [synth.make_spectrum(n, 6, seed=100 + k % 8) for k, n in enumerate(lengths)]
                best_dt = None
                for _ in range(2):
                    tb = time.perf_counter()
                    with FitBatch([(q["w"], q["u"], q["v"], q["weights"]) for q in sps], [q["lower"] for q in sps],
                                  [q["upper"] for q in sps], swarmsize=204, seeds=list(range(7, 7 + Kb)), device=device,
                                  minstep=-1.0, minfunc=-1.0) as fb:
                        fb.run(2000, 64)
                        fb.status()
                    dtb = time.perf_counter() - tb
                    best_dt = dtb if best_dt is None else min(best_dt, dtb)
                rag[name] = {"wall_ms": best_dt * 1e3, "fits_per_s": Kb / best_dt,
                             "units_per_s": 204.0 * float(np.sum(lengths)) * 6 * 2001 / best_dt}
            rag["ragged_over_equal_fits_per_s"] = rag["ragged"]["fits_per_s"] / rag["equal"]["fits_per_s"]
            rag["note"] = ("%d default-size swarms, 2000 generations each (stopping rule off), best of two runs: spectra of "
                           "different lengths share a batch in the wave = particle geometry (a wave reads its fit's length "
                           "from the fit's record); against equal lengths at the 512-multiple next to the mean" % Kb)
            batched_fit["ragged_lengths"] = rag
            # ... and the user-level call, host side included: nmrfit_amd.fit_many on the same 40 spectra (weights and plans on
            # the host, one device batch, results into FitUtility objects)
            import contextlib
            import io
            import nmrfit_amd
            jobs_b = [(synth.SynthData(q["w"], q["u"], q["v"], q["peaks"]), list(q["lower"]), list(q["upper"])) for q in specs]
            e2e = {}
            for key, rule in (("stopping_rule_off", {"minstep": -1.0, "minfunc": -1.0}), ("stopping_rule_on", {})):
                with contextlib.redirect_stdout(io.StringIO()):
                    tb = time.perf_counter()
                    res = nmrfit_amd.fit_many([dict(data=j[0], lower=j[1], upper=j[2], options=dict(rule, seed=7 + k, device=device))
                                               for k, j in enumerate(jobs_b)])
                    dtb = time.perf_counter() - tb
                e2e[key] = {"wall_ms": dtb * 1e3, "fits_per_s": Kb / dtb, "error_fit0": float(res[0].error)}
            batched_fit["fit_many_end_to_end"] = e2e
            batched_fit["note"] = ("nmrfit_amd.batch.FitBatch: context creation + generation 0 + generations + read-back of "
                                   "%d fits of the reference's default shape in ONE batch (what nmrfit_amd.fit_many builds); "
                                   "stopping_rule_off runs all 2000 generations of every fit (the work of "
                                   "reference_default_fit x %d), stopping_rule_on is pyswarm's rule with its defaults "
                                   "(1e-8): every fit stops on its own.  Ceiling of the rule-off case by instruction count: "
                                   "~4500 fp64 VALU instructions per particle and generation (profiles/r05/batch_instr_model.txt) "
                                   "-> ~330 fits/s if every issue slot of a 2.4 GHz clock were used; the kernel holds ~0.8 "
                                   "of that, like the headline kernel" % (Kb, Kb))
        except Exception as e:      # reported, never fatal for the headline
            batched_fit = {"error": repr(e)}
    # ... and the REST of the reference's per-spectrum script (README.md:64-72): fit -> generate_result ->
    # calculate_area_fraction (nmrfit/utils.py:226-295, 297-322).  Round 6: the reconstruction of every fit of a device
    # batch is one launch over the batch's resident spectra (nmrfit_batch_contributions, csrc/result.hip) and the batches
    # go through a three-stage pipeline (prepare | run | read back): fit_many(jobs, generate=True) against the plain loop
    readme_pipeline = None
    if rank == 0 and world == 1 and args.workload == "C3" and args.variant == 0 and args.other_configs:
        try:
            import contextlib
            import io
            import nmrfit_amd
            Kp, Lp = 200, 12
            specs_p = [synth.make_spectrum(4096, 6, seed=100 + k % 8) for k in range(8)]

            def jobs_p(n, rule):
                return [dict(data=synth.SynthData(q["w"], q["u"], q["v"], q["peaks"]), lower=list(q["lower"]),
                             upper=list(q["upper"]), options=dict(rule, seed=7 + k, device=device))
                        for k, q in ((k, specs_p[k % 8]) for k in range(n))]
            readme_pipeline = {"shape": {"jobs": Kp, "swarm": 204, "grid": 4096, "peaks": 6, "loop_jobs": Lp}}
            with contextlib.redirect_stdout(io.StringIO()):
                nmrfit_amd.fit_many(jobs_p(8, {"maxiter": 5}), generate=True)      # (the reconstruction kernel's code object loads now)
            for key, rule in (("stopping_rule_off", {"minstep": -1.0, "minfunc": -1.0}), ("stopping_rule_on", {})):
                with contextlib.redirect_stdout(io.StringIO()):
                    dt_fit = dt_full = None
                    for _ in range(2):      # (best of two: the first run of each kind also pays for fresh host memory)
                        t_a = time.perf_counter()
                        nmrfit_amd.fit_many(jobs_p(Kp, rule))
                        t_b = time.perf_counter()
                        full = nmrfit_amd.fit_many(jobs_p(Kp, rule), generate=True)
                        fracs = [f.calculate_area_fraction() for f in full]
                        t_c = time.perf_counter()
                        dt_fit = (t_b - t_a) if dt_fit is None else min(dt_fit, t_b - t_a)
                        dt_full = (t_c - t_b) if dt_full is None else min(dt_full, t_c - t_b)
                    t_a, t_b = 0.0, dt_fit
                    t_c = t_b + dt_full
                    t_loop0 = time.perf_counter()
                    loop = []
                    for j in jobs_p(Lp, rule):
                        f1 = nmrfit_amd.fit(j["data"], j["lower"], j["upper"], summary=False, options=j["options"])
                        f1.generate_result()
                        loop.append((f1, f1.calculate_area_fraction()))
                    t_d = t_c + (time.perf_counter() - t_loop0)
                same = all(np.array_equal(a.params, b.params) and np.array_equal(a.u, b.u) and np.array_equal(a.V, b.V)
                           and np.array_equal(a.imag_contribs[-1], b.imag_contribs[-1]) and fr == fb
                           for a, (b, fb), fr in zip(full, loop, fracs))
                readme_pipeline[key] = {
                    "fit_only_fits_per_s": Kp / (t_b - t_a), "pipeline_fits_per_s": Kp / (t_c - t_b),
                    "pipeline_over_fit_only": (t_b - t_a) / (t_c - t_b),
                    "plain_loop_ms_per_spectrum": (t_d - t_c) / Lp * 1e3, "plain_loop_fits_per_s": Lp / (t_d - t_c),
                    "batched_equals_plain_loop_bit_for_bit": bool(same),
                    "result_bytes_per_fit": int(sum(a.nbytes for a in (full[0].u, full[0].v, full[0].V, full[0].I,
                                                                        full[0].data.V, full[0].data.I))
                                                + sum(a.nbytes for a in full[0].real_contribs + full[0].imag_contribs))}
            readme_pipeline["note"] = (
                "fit -> generate_result -> calculate_area_fraction per spectrum (README.md:64-72) for %d default-shape "
                "spectra: nmrfit_amd.fit_many(jobs, generate=True) (device batches of <= 64 fits; per batch ONE "
                "reconstruction launch over its resident spectra, results through a pinned double buffer into numpy arrays) "
                "against fit_many without the reconstruction (best of two runs each) and against the plain loop over "
                "nmrfit_amd.fit + generate_result (%d spectra); `pipeline_over_fit_only` is the bar of VERDICT r5 (>= 0.8)" % (Kp, Lp))
        except Exception as e:      # reported, never fatal for the headline
            readme_pipeline = {"error": repr(e)}
    # N > 1, the OTHER multi-GPU mode: spectra-parallel replicas (nmrfit_amd.fit_many(shard=True)).  `value`
    # above is the swarm-sharded C4 case; the reference's own workload -- many 204-particle fits, nmrfit/utils.py:177,
    # looped over spectra, README.md:64-66 -- scales as replicas: every rank fits its own spectra as device batches on its
    # own GPU and NOTHING crosses ranks but the result records at the end.  Every rank takes part (the calls are
    # collective over the rendezvous channel).
    replicas = None
    if (use_dist and channel is not None and args.variant == 0 and not args.no_extras
            and os.environ.get("NMRFIT_BENCH_NO_REPLICAS") != "1"):
        try:
            import contextlib
            import io
            import nmrfit_amd
            Kr = int(os.environ.get("NMRFIT_BENCH_REPLICA_JOBS", "200"))
            specs_r = [synth.make_spectrum(4096, 6, seed=100 + k % 8) for k in range(8)]

            def jobs_r(indices, rule):
                return [dict(data=synth.SynthData(q["w"], q["u"], q["v"], q["peaks"]), lower=list(q["lower"]),
                             upper=list(q["upper"]), options=dict(rule, seed=7 + k, device=device))
                        for k, q in ((k, specs_r[k % 8]) for k in indices)]
            with contextlib.redirect_stdout(io.StringIO()):
                nmrfit_amd.fit_many(jobs_r(range(8), {"maxiter": 5}))        # (the batch kernels' code objects load now)
            replicas = {"jobs_per_rank": Kr, "shape": {"swarm": 204, "grid": 4096, "peaks": 6, "generations": 2000},
                        "ranks": world}
            for key, rule in (("stopping_rule_on", {}), ("stopping_rule_off", {"minstep": -1.0, "minfunc": -1.0})):
                # (1) every rank its own Kr spectra, nothing exchanged: the replicas' own rates
                barrier()
                with contextlib.redirect_stdout(io.StringIO()):
                    t_a = time.perf_counter()
                    mine_r = nmrfit_amd.fit_many(jobs_r(range(rank * Kr, (rank + 1) * Kr), rule))
                    dt_own = time.perf_counter() - t_a
                dts = [json.loads(b.decode()) for b in channel.all_gather(json.dumps(dt_own).encode())]
                # (2) the user-level call: ONE list of world x Kr jobs, divided over the ranks, results gathered everywhere
                barrier()
                with contextlib.redirect_stdout(io.StringIO()):
                    t_a = time.perf_counter()
                    all_r = nmrfit_amd.fit_many(jobs_r(range(world * Kr), rule), shard=True, channel=channel)
                    dt_sh = time.perf_counter() - t_a
                dts_sh = [json.loads(b.decode()) for b in channel.all_gather(json.dumps(dt_sh).encode())]
                same = all(np.array_equal(all_r[rank + world * i].params, mine_r[rank + world * i - rank * Kr].params)
                           for i in range(Kr) if rank * Kr <= rank + world * i < (rank + 1) * Kr)
                rates = [Kr / d for d in dts]
                replicas[key] = {
                    "per_rank_fits_per_s": rates, "per_rank_spread": {"min": min(rates), "max": max(rates)},
                    "aggregate_fits_per_s": world * Kr / max(dts),
                    "expected_aggregate_fits_per_s": float(np.sum(rates)),
                    "aggregate_over_expected": (world * Kr / max(dts)) / float(np.sum(rates)),
                    "sharded_call": {"aggregate_fits_per_s": world * Kr / max(dts_sh), "wall_ms_max_rank": max(dts_sh) * 1e3,
                                     "results_gathered_on_every_rank": len(all_r) == world * Kr and all(f is not None for f in all_r),
                                     "own_share_equals_local_run": bool(same)}}
            replicas["note"] = (
                "spectra-parallel replicas: every rank fits its own %d default-shape spectra as device batches on its own GPU "
                "(per_rank_fits_per_s; aggregate = all spectra / the slowest rank's time; expected = the sum of the ranks' own "
                "rates: no collective, so anything below 1.0 is rank imbalance, i.e. the clocks the GPUs hold), and the "
                "user-level call nmrfit_amd.fit_many(jobs, shard=True) over ONE list of %d jobs, the (params, error, seed) "
                "records gathered on every rank over the rendezvous sockets.  The replica prediction for N GPUs is N x the "
                "single-GPU rate (`reference_default_fit_batched.fit_many_end_to_end` of an N = 1 run)" % (Kr, world * Kr))
        except Exception as e:      # reported, never fatal for the headline
            replicas = {"error": repr(e)}
            extras_errors.append("replicas: %r" % (e,))
    # the host-pointer entry point (X uploaded, f downloaded every call): the PCIe-inclusive
    # rate, reported beside the resident one -- never as `value`
    try:
        if rank == 0 and world == 1 and not args.no_extras:
            Xh = sw.state()["x"]
            for _ in range(3):
                ev.objective_batch(Xh)
            host_all = []
            for _ in range(20):
                t1 = time.perf_counter()
                ev.objective_batch(Xh)
                host_all.append((time.perf_counter() - t1) * 1e3)
            host_ms = float(np.mean(host_all))
            host_min_ms = float(np.min(host_all))
    except Exception as e:      # an extra must never cost the run its headline line
        extras_errors.append("host_pointer_call: %r" % (e,))
        host_ms = None

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
        traffic_live = False
        l2_hit = None
        mem = dict((pmc_live or {}).get("fetch", {}).get("counters", {}))
        mem.update((pmc_live or {}).get("write", {}).get("counters", {}))
        if "FETCH_SIZE" in mem and "WRITE_SIZE" in mem:
            # MI355X_MICROARCH.md (HBM): both in KiB; on gfx950 FETCH_SIZE reads 1/2 of the bytes of wide
            # coalesced reads -> doubled; WRITE_SIZE is exact
            traffic = 2.0 * mem["FETCH_SIZE"] * 1024.0 + mem["WRITE_SIZE"] * 1024.0
            traffic_live = True
            if mem.get("TCC_HIT_sum") is not None and mem.get("TCC_MISS_sum") is not None:
                l2_hit = mem["TCC_HIT_sum"] / max(1.0, mem["TCC_HIT_sum"] + mem["TCC_MISS_sum"])
        pmcf = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        if traffic is None and os.path.exists(pmcf):
            try:
                traffic = json.load(open(pmcf)).get(args.workload, {}).get("hbm_bytes_per_launch")
            except Exception:
                traffic = None
        valu = {"bound": "fp64_valu_issue", "unit": "wave64 fp64 VALU instructions/s (peak: 1024 SIMDs x 2.4 GHz / 4 cycles "
                                                     "per instruction; frac = achieved / peak)"}
        summ = next((p for p in PMC_SUMMARIES if os.path.exists(os.path.join(ROOT, p))), None)
        sq = (pmc_live or {}).get("sq", {})
        sqc = sq.get("counters", {})
        if all(k in sqc for k in ("SQ_INSTS_VALU", "SQ_ACTIVE_INST_VALU", "SQ_BUSY_CYCLES")):
            # counters of THIS run's child pass (same build, same box, minutes apart)
            insts = sqc["SQ_INSTS_VALU"]                                   # wave-instructions per launch
            ipu = insts * 64.0 / units_launch
            peak = SIMDS * PEAK_CLOCK_MHZ * 1e6 / FP64_ISSUE_CYCLES
            achieved = insts / (t_kernel_ms * 1e-3)
            kernel_cycles = sqc["SQ_BUSY_CYCLES"] / 32.0                   # summed over 32 shader engines
            valu.update({"achieved": achieved, "peak": peak, "frac": achieved / peak,
                         "achieved_unit": "fp64 wave64 VALU instructions/s",
                         "valu_instructions_per_unit": ipu,
                         "valu_busy_frac": 4.0 * sqc["SQ_ACTIVE_INST_VALU"] / SIMDS / kernel_cycles,
                         "valu_cycles_per_instruction": 4.0 * sqc["SQ_ACTIVE_INST_VALU"] / insts,
                         "from_committed_profile": False,
                         "pmc_source": "live: child pass of this run under rocprofv3 --kernel-trace --pmc (%d launches, "
                                       "kernel %.4f ms in the child)" % (sq.get("launches", 0), sq.get("kernel_ms_in_child") or -1),
                         "note": "instruction count, busy fraction and cycles per instruction from this run's own "
                                 "rocprofv3 child pass; `achieved` divides that count by the kernel time of the "
                                 "timed region above"})
        elif args.workload == "C3" and args.variant == 0 and summ:
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
                       "spectrum": "sparse lines (SURVEY 8(d) synthetic: widths 0.4-0.6 % of the span; see "
                                   "`dense_spectrum` for the other end of the range)",
                       "exchange": exchange_desc, "variant": args.variant, "generations_done": st["iteration"],
                       "swarm_best_f": st["fg"], "preheat_launches": heat_launches},
            "step_ms": sst, "kernel_ms": kst,
            # the resource that BINDS this kernel: fp64 vector-ALU issue.  achieved = wave64 VALU
            # instructions per second (this run's SQ_INSTS_VALU per launch / this run's kernel time), peak = 1024 SIMDs x
            # 2.4 GHz / 4 cycles per fp64 instruction; `traffic` = physical HBM bytes per launch (PMC)
            "roofline": dict(valu, kernel="objective_kernel", kernel_ms=t_kernel_ms,
                             kernel_ms_note="mean of the HIP-event durations of the objective kernel alone, one pair "
                                            "per timed step, on the stream it is launched on",
                             units_per_launch=units_launch, launch=geom, traffic=traffic,
                             traffic_from_committed_profile=traffic is not None and not traffic_live,
                             traffic_source=("live: child passes of this run under rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE "
                                             "(2 x FETCH_SIZE KiB + WRITE_SIZE KiB per launch, the gfx950 correction)"
                                             if traffic_live else "profiles/pmc_traffic.json"),
                             l2_hit_rate=l2_hit),
            # the roofline the METRIC names (BASELINE.json: "HBM GB/s vs peak"), under SURVEY 8(d)(i)'s streaming-operand
            # model: an EFFECTIVE rate -- the operands are L2-resident, physical HBM traffic is `roofline.traffic`
            "roofline_streaming": {"bound": "hbm", "rate_kind": "effective",
                                   "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS,
                                   # also against the 6.29 TB/s a float4 copy measures on this part; > 1 here IS the
                                   # evidence that the streaming figure is served from L2, not from HBM
                                   "peak_measured_copy": HBM_COPY_GBS, "frac_vs_measured_copy_peak": ach / HBM_COPY_GBS,
                                   "physical_hbm_frac_of_peak": (traffic / (t_kernel_ms * 1e-3) / 1e9 / HBM_PEAK_GBS
                                                                 if traffic else None),
                                   "model": "streaming-operand bytes S*(4*N*8)+S*D*8+S*8 per launch (SURVEY 8(d)(i)); "
                                            "w/u/v/weights are shared by all particles and L2-resident, so this is an "
                                            "effective rate, not physical HBM traffic (`roofline.traffic`); the binding "
                                            "resource is fp64 VALU issue: `roofline`",
                                   "bytes_per_launch": bytes_launch, "kernel_ms": t_kernel_ms},
        }
        if kst and not (t_kernel_ms <= ms_per_step * 1.0005):
            line["error"] = "objective kernel (%.4f ms) longer than the step that contains it (%.4f ms)" % (
                t_kernel_ms, ms_per_step)
            rc = 3
        if world > 1 or use_dist:
            # how many ranks RCCL itself saw (null: the exchange that ran was not RCCL)
            line["rccl"] = rccl_info
        if world > 1 and rccl_info is None and not host_requested:
            line["error"] = ("N = %d did not run over RCCL (%s): `value` was measured with the candidate record "
                             "staged through the host and is NOT an RCCL scaling number" % (world, rccl_failure))
            rc = 5
        elif world > 1 and rccl_info is not None and rccl_info["ranks_counted_by_all_reduce"] != world:
            line["error"] = "RCCL counted %d ranks, expected %d" % (rccl_info["ranks_counted_by_all_reduce"], world)
            rc = 5
        # What this line should read on N GPUs if the sharding costs what its parts cost (so that the first measured
        # scaling curve can be judged against a prediction made BEFORE it): every rank runs the same 4096-particle
        # shard, so a generation is the slowest rank's objective kernel + what a single-rank generation adds to its
        # kernel (candidate / fold launches: measured here at N = 1) + one all-gather of (D + 1) doubles per rank +
        # the fold launch that follows it.  The all-gather is latency-bound (616 B per rank at D = 76): 15-40 us on
        # xGMI is assumed until measured.
        overhead_ms = (ms_per_step - t_kernel_ms) if world == 1 else None
        ag_lo, ag_hi, fold_ms = 0.015, 0.040, 0.006
        kern = float(ranks["kernel_ms_mean"]["max"]) if ranks else t_kernel_ms
        base_over = overhead_ms if overhead_ms is not None else 0.015     # (N = 1 of this build: ~0.015 ms)
        line["scaling_model"] = {
            "model": "ms_per_step(N) = max_rank(kernel_ms) + single_rank_overhead_ms + all_gather_ms + fold_launch_ms, "
                     "N > 1; weak scaling: value(N) = N * units_per_launch / ms_per_step(N)",
            "single_rank_overhead_ms": base_over, "all_gather_ms_assumed": [ag_lo, ag_hi], "fold_launch_ms": fold_ms,
            "max_rank_kernel_ms": kern,
            "expected_ms_per_step": ([kern + base_over + ag_lo + fold_ms, kern + base_over + ag_hi + fold_ms]
                                     if world > 1 else [ms_per_step, ms_per_step]),
            "expected_weak_scaling_efficiency": [kern_eff for kern_eff in (
                (t_kernel_ms + base_over) / (kern + base_over + ag_hi + fold_ms),
                (t_kernel_ms + base_over) / (kern + base_over + ag_lo + fold_ms))] if world > 1 else [1.0, 1.0],
            "measured_over_expected": ([ms_per_step / (kern + base_over + ag_hi + fold_ms),
                                        ms_per_step / (kern + base_over + ag_lo + fold_ms)] if world > 1 else None),
            "replicas": {"model": "fits_per_s(N) = N x single-GPU fits_per_s: the ranks fit different spectra and exchange "
                                  "nothing during the fits (nmrfit_amd.fit_many(shard=True)); measured in `replicas`",
                         "expected_aggregate_fits_per_s": ((replicas or {}).get("stopping_rule_on") or {}).get("expected_aggregate_fits_per_s"),
                         "measured_aggregate_fits_per_s": ((replicas or {}).get("stopping_rule_on") or {}).get("aggregate_fits_per_s")},
            "note": "at C3 / C4 size the exchange is 2-4 % of a generation: near-linear weak scaling is expected, and a "
                    "rank that holds a lower clock at its power cap (kernel_ms spread in `ranks`) costs more than the "
                    "collective.  Small swarms are the opposite: see nmrfit_amd.utils.small_shard_warning and "
                    "fit_many(shard=True)"}
        line["power"] = power.summary()      # (null where the hwmon files are not readable)
        if ranks is not None:
            line["ranks"] = ranks
        if variants is not None:
            line["variants"] = variants
            line["farfield_variant"] = farfield
            # the kernel nmrfit_amd.fit() itself picks at this problem size (utils.default_variant):
            # measured here beside the headline, never `value`
            from nmrfit_amd.utils import default_variant
            dv = default_variant(N, P)
            fd = {"variant": dv, "kernel_ms": variants[dv + "_ms"],
                  "units_per_s": units_launch / (variants[dv + "_ms"] * 1e-3),
                  "max_rel_diff_vs_default": variants[dv + "_max_rel_diff_vs_default"],
                  "note": "what fit() runs when options['variant'] is absent at this grid x peaks; kernel alone, "
                          "this run"}
            fsum = next((q for q in FARFIELD_PMC_SUMMARIES if os.path.exists(os.path.join(ROOT, q))), None)
            ffc = (pmc_live or {}).get("sq_farfield", {}).get("counters", {})
            if dv == "farfield" and all(k in ffc for k in ("SQ_INSTS_VALU", "SQ_ACTIVE_INST_VALU", "SQ_BUSY_CYCLES")):
                fd.update({"valu_instructions_per_unit": ffc["SQ_INSTS_VALU"] * 64.0 / units_launch,
                           "valu_busy": 4.0 * ffc["SQ_ACTIVE_INST_VALU"] / SIMDS / (ffc["SQ_BUSY_CYCLES"] / 32.0),
                           "valu_cycles_per_instruction": 4.0 * ffc["SQ_ACTIVE_INST_VALU"] / ffc["SQ_INSTS_VALU"],
                           "from_committed_profile": False,
                           "pmc_source": "live: child pass of this run under rocprofv3 --pmc (--variant 6)"})
            elif dv == "farfield" and args.workload == "C3" and fsum:
                try:
                    fm = json.load(open(os.path.join(ROOT, fsum)))
                    fi = fm["objective_kernel_pmc"]["SQ_INSTS_VALU"]["mean"]
                    fd.update({"valu_instructions_per_unit": fi * 64.0 / (4096.0 * 65536.0 * 24.0),
                               "valu_busy": fm.get("valu_busy_frac"),
                               "valu_cycles_per_instruction": fm.get("valu_cycles_per_inst"),
                               "from_committed_profile": True, "pmc_source": fsum})
                except Exception as e:
                    fd["pmc_error"] = repr(e)
            line["fit_default"] = fd
            # the opt-in mixed-precision form of that kernel: never `value`, never selected automatically
            line["mixed_precision"] = {
                "variant": "farfield32", "kernel_ms": variants["farfield32_ms"],
                "units_per_s": units_launch / (variants["farfield32_ms"] * 1e-3),
                "kernel_ms_fp64_farfield": variants["farfield_ms"],
                "speedup_vs_fp64_farfield": variants["farfield_ms"] / variants["farfield32_ms"],
                "max_rel_diff_vs_default": variants["farfield32_max_rel_diff_vs_default"],
                "note": "orders 1..15 of the far-field kernel's shared polynomial in packed fp32 (v_pk_fma_f32), everything "
                        "else fp64; options={'variant': 'farfield32'}"}
        def counters_of(name, units, kernel_ms):
            """instructions per unit, busy and issue fraction of an extra PMC pass (objective launches only)"""
            c = (pmc_live or {}).get(name, {})
            cc = c.get("counters", {})
            if not all(k in cc for k in ("SQ_INSTS_VALU", "SQ_ACTIVE_INST_VALU", "SQ_BUSY_CYCLES")):
                return None
            kms = kernel_ms or c.get("kernel_ms_in_child")
            return {"valu_instructions_per_unit": cc["SQ_INSTS_VALU"] * 64.0 / units,
                    "valu_busy_frac": 4.0 * cc["SQ_ACTIVE_INST_VALU"] / SIMDS / (cc["SQ_BUSY_CYCLES"] / 32.0),
                    "frac": (cc["SQ_INSTS_VALU"] / (kms * 1e-3)) / (SIMDS * PEAK_CLOCK_MHZ * 1e6 / FP64_ISSUE_CYCLES) if kms else None,
                    "kernel_ms_in_pmc_child": c.get("kernel_ms_in_child"), "launches": c.get("launches"),
                    "pmc_source": "live: child pass of this run under rocprofv3 --kernel-trace --pmc (bench.py --kernel-only)"}
        if variants is not None and "noskip_ms" in variants:
            line["every_unit"] = {"kernel": "objective_kernel<NOSKIP> (A/B library)", "kernel_ms": variants["noskip_ms"],
                                  "units_per_s": units_launch / (variants["noskip_ms"] * 1e-3),
                                  "roofline": counters_of("sq_noskip", units_launch, None),
                                  "note": "every (particle, point, peak) unit evaluated: no Gaussian window skip"}
        if imag is not None:
            line["imaginary_channel"] = imag
        if dense is not None:
            dense["roofline"] = counters_of("sq_dense", units_launch, None)
            dense["farfield"]["roofline"] = counters_of("sq_dense_farfield", units_launch, None)
            line["dense_spectrum"] = dense
        if others:
            line["other_configs"] = others
        if default_fit is not None:
            line["reference_default_fit"] = default_fit
        if batched_fit is not None:
            line["reference_default_fit_batched"] = batched_fit
        if readme_pipeline is not None:
            line["readme_pipeline"] = readme_pipeline
        if replicas is not None:
            line["replicas"] = replicas
        if host_ms is not None:
            # SURVEY 8(d) words the metric's t_generation "incl. H2D of X and D2H of f": that rate, beside `value` (resident
            # state: what a fit runs on -- the swarm never leaves the device)
            line["value_incl_pcie"] = units_launch / (host_ms * 1e-3)
            line["host_pointer_call"] = {"ms": host_ms, "units_per_s": units_launch / (host_ms * 1e-3),
                                         "over_resident_kernel": host_ms / t_kernel_ms if t_kernel_ms == t_kernel_ms else None,
                                         "ms_min": locals().get("host_min_ms"),
                                         "note": "nmrfit_objective_batch with host X / f (pageable numpy arrays), mean of 20 calls: "
                                                 "H2D of X + kernel + D2H of f; what a third-party optimiser that keeps its swarm "
                                                 "on the host pays per generation (nmrfit/utils.py:176).  `value_incl_pcie` is this "
                                                 "rate.  A sliced upload overlapping the kernels was measured and rejected "
                                                 "(profiles/r06/host_pointer_pipelined_ab.txt)"}
        if extras_errors:
            line["extras_errors"] = extras_errors
        if pmc_live is not None and pmc_live.get("errors"):
            line["pmc_live_errors"] = pmc_live["errors"]
        if cpu is not None:
            line["cpu_baseline"] = cpu
        sys.stdout.flush()
        os.write(json_fd, (json.dumps(line) + "\n").encode())
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
