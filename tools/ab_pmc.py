#!/usr/bin/env python3
"""Counters per build: launches each library's objective kernel REPS times, one library after the other,
so that a `rocprofv3 --kernel-trace --pmc ...` pass over this script yields the counters of every build in
dispatch order; `--summarise DIR` then groups the objective_kernel rows of the counter CSV in runs of REPS.
    rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY \
        SQ_WAIT_INST_ANY --output-format csv -d OUT -- python3 tools/ab_pmc.py --variant 6 lib1.so lib2.so ...
    python3 tools/ab_pmc.py --summarise OUT lib1.so lib2.so ..."""
import argparse, collections, csv, ctypes, glob, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
REPS = 4


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("libs", nargs="+")
    ap.add_argument("--variant", type=int, default=6)
    ap.add_argument("--workload", default="C3")
    ap.add_argument("--summarise", default=None)
    a = ap.parse_args()
    if a.summarise:
        fs = glob.glob(os.path.join(a.summarise, "**", "*_counter_collection.csv"), recursive=True)
        rows = collections.defaultdict(dict)      # dispatch id -> counter -> value
        for r in csv.DictReader(open(fs[0])):
            if "objective_kernel" in r["Kernel_Name"]:
                rows[int(r["Dispatch_Id"])][r["Counter_Name"]] = float(r["Counter_Value"])
        ids = sorted(rows)
        assert len(ids) == REPS * len(a.libs), (len(ids), len(a.libs))
        from nmrfit_amd import synth
        cfg = synth.CONFIGS[a.workload]
        chunks = cfg.S * (cfg.N // 512)
        print("%-22s %10s %10s %8s %8s %9s %9s %9s" % ("build", "valu/chunk", "cyc/chunk", "busy", "cyc/inst", "parked", "issuewait", "wavecyc"))
        for i, lib in enumerate(a.libs):
            grp = [rows[j] for j in ids[i * REPS + 1:(i + 1) * REPS]]      # skip each build's first launch
            m = {k: sum(g[k] for g in grp) / len(grp) for k in grp[0]}
            kc = m["SQ_BUSY_CYCLES"] / 32.0
            print("%-22s %10.1f %10.0f %8.3f %8.2f %9.0f %9.0f %9.0f" % (
                os.path.basename(lib), m["SQ_INSTS_VALU"] / chunks, kc * 1024 / chunks * 1.0,
                4 * m["SQ_ACTIVE_INST_VALU"] / 1024 / kc, 4 * m["SQ_ACTIVE_INST_VALU"] / m["SQ_INSTS_VALU"],
                4 * m["SQ_WAIT_ANY"] / chunks, 4 * m["SQ_WAIT_INST_ANY"] / chunks, 4 * m["SQ_WAVE_CYCLES"] / chunks))
        return
    from nmrfit_amd import _cabi, synth
    from nmrfit_amd.equations import Evaluator
    cfg = synth.CONFIGS[a.workload]
    sp = synth.make_spectrum(cfg.N, cfg.P, seed=1)
    X = synth.make_swarm(sp["lower"], sp["upper"], cfg.S, seed=2, x_true=sp["x_true"])
    for path in a.libs:
        L = ctypes.CDLL(os.path.abspath(path))
        for name, argtypes in _cabi.ALL_SIGNATURES.items():
            if not hasattr(L, name): continue      # (an older build of the library: entry points added since)
            fn = getattr(L, name); fn.argtypes = argtypes; fn.restype = ctypes.c_int
        L.nmrfit_last_error.argtypes = []; L.nmrfit_last_error.restype = ctypes.c_char_p
        _cabi._LIB = L
        with Evaluator(sp["w"], sp["u"], sp["v"], sp["weights"]) as ev:
            ev.set_variant(a.variant)
            dX = ev.dev_alloc(X.nbytes); df = ev.dev_alloc(8 * cfg.S)
            ev.upload(dX, X)
            for _ in range(REPS):
                ev.objective_batch_dev(cfg.S, cfg.P, dX, df)
            ev.synchronize()
            ev.dev_free(dX); ev.dev_free(df)


if __name__ == "__main__":
    main()
