#!/usr/bin/env python3
"""Summarise a tools/profile.sh (or tools/pmc_cmd.sh) output directory: per-kernel stats + mean PMC counters of one
kernel (default: the objective kernel).
Usage: tools/pmc_summary.py gpurun_out/prof/<tag> [--kernel <substring of its name>] [--json out.json]"""
import collections
import csv
import glob
import json
import os
import sys


def main():
    d = sys.argv[1]
    kname = sys.argv[sys.argv.index("--kernel") + 1] if "--kernel" in sys.argv else "objective_kernel"
    out = {"kernel": kname}
    def total_ns(path):
        return sum(float(r["TotalDurationNs"]) for r in csv.DictReader(open(path)))
    # (newest run of this tag; a bench run's child processes write stats files of their own -- the main process ran longest)
    st = sorted(glob.glob(os.path.join(d, "trace", "*", "*_kernel_stats.csv")), key=os.path.getmtime)
    if st:
        newest = os.path.getmtime(st[-1])
        st = [max((f for f in st if newest - os.path.getmtime(f) < 600), key=total_ns)]
        rows = list(csv.DictReader(open(st[-1])))
        def short(n):      # "void nmrfit::(anonymous namespace)::objective_kernel<0, false, 0, 4>(double const*, ...)"
            n = n.replace("nmrfit::(anonymous namespace)::", "").replace("void ", "")
            return n.split("(")[0][-60:]
        out["kernel_stats"] = [{"name": short(r["Name"]), "calls": int(r["Calls"]),
                                "avg_ns": float(r["AverageNs"]), "min_ns": float(r["MinNs"]),
                                "max_ns": float(r["MaxNs"]), "pct": float(r["Percentage"])} for r in rows[:8]]
    pmc = {}
    for sub in ("pmc_sq", "pmc_fetch", "pmc_write", "pmc_lds"):
        fs = sorted(glob.glob(os.path.join(d, sub, "*", "*_counter_collection.csv")), key=os.path.getmtime)
        if not fs:
            continue
        agg = collections.defaultdict(list)
        for r in csv.DictReader(open(fs[-1])):
            if kname in r["Kernel_Name"]:
                agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, v in agg.items():
            pmc[k] = {"mean": sum(v) / len(v), "n": len(v)}
    out["objective_kernel_pmc"] = pmc
    if "FETCH_SIZE" in pmc:
        # MI355X_MICROARCH.md (HBM): FETCH_SIZE is in KiB and reads 1/2 of the bytes of wide
        # coalesced reads on gfx950 -> double it; WRITE_SIZE is exact (KiB).
        fetch = pmc["FETCH_SIZE"]["mean"] * 1024 * 2
        write = pmc.get("WRITE_SIZE", {"mean": 0})["mean"] * 1024
        out["hbm_bytes_per_launch"] = fetch + write
        out["hbm_bytes_note"] = "2*FETCH_SIZE*1024 + WRITE_SIZE*1024 (gfx950 correction per MI355X_MICROARCH.md)"
    if "GRBM_GUI_ACTIVE" in pmc and st:
        ks = [k for k in out["kernel_stats"] if kname in k["name"]]
        if ks:
            out["clock_ghz_est"] = pmc["GRBM_GUI_ACTIVE"]["mean"] / 8 / ks[0]["avg_ns"]
    if "SQ_INSTS_VALU" in pmc and "SQ_ACTIVE_INST_VALU" in pmc:
        out["valu_cycles_per_inst"] = 4 * pmc["SQ_ACTIVE_INST_VALU"]["mean"] / pmc["SQ_INSTS_VALU"]["mean"]
    if "SQ_BUSY_CYCLES" in pmc and "SQ_ACTIVE_INST_VALU" in pmc:
        # SQ_BUSY_CYCLES sums 32 shader engines; SQ_ACTIVE_INST_VALU counts quad-cycles over 1024 SIMDs
        kernel_cycles = pmc["SQ_BUSY_CYCLES"]["mean"] / 32
        out["kernel_cycles"] = kernel_cycles
        out["valu_busy_frac"] = 4 * pmc["SQ_ACTIVE_INST_VALU"]["mean"] / 1024 / kernel_cycles
    if "SQ_INSTS_VALU" in pmc and st:
        ks = [k for k in out["kernel_stats"] if kname in k["name"]]
        if ks:   # wave-instructions issued per SIMD and second, against 2.4 GHz / 4 cycles per fp64 instruction
            out["valu_issue_frac_of_2p4GHz"] = pmc["SQ_INSTS_VALU"]["mean"] * 4 / 1024 / (ks[0]["avg_ns"] * 2.4)
    print(json.dumps(out, indent=1))
    if "--json" in sys.argv:
        json.dump(out, open(sys.argv[sys.argv.index("--json") + 1], "w"), indent=1)


if __name__ == "__main__":
    main()
