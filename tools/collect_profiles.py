#!/usr/bin/env python3
"""Copy the judged summaries of a tools/profile.sh run into profiles/<round>/:
   tools/collect_profiles.py gpurun_out/prof/<tag> profiles/r01 [prefix, default bench_c3]"""
import csv
import glob
import json
import os
import shutil
import subprocess
import sys

src, dst = sys.argv[1], sys.argv[2]
prefix = sys.argv[3] if len(sys.argv) > 3 else "bench_c3"
os.makedirs(dst, exist_ok=True)
def total_ns(path):     # (the bench's child processes write stats files of their own: the main process ran longest)
    return sum(float(r["TotalDurationNs"]) for r in csv.DictReader(open(path)))


stats = max(glob.glob(os.path.join(src, "trace", "*", "*_kernel_stats.csv")), key=total_ns)
shutil.copy(stats, os.path.join(dst, prefix + "_kernel_stats.csv"))
keep = ["Dispatch_Id", "Grid_Size", "Workgroup_Size", "LDS_Block_Size", "VGPR_Count", "Accum_VGPR_Count", "SGPR_Count",
        "Counter_Name", "Counter_Value"]
for sub in ("sq", "fetch", "write", "lds"):
    fs = sorted(glob.glob(os.path.join(src, "pmc_" + sub, "*", "*_counter_collection.csv")), key=os.path.getmtime)
    if not fs:
        continue
    with open(fs[-1]) as f, open(os.path.join(dst, "%s_pmc_%s_objective_kernel.csv" % (prefix, sub)), "w", newline="") as o:
        w = csv.writer(o)
        w.writerow(keep)
        for r in csv.DictReader(f):
            if "objective_kernel" in r["Kernel_Name"]:
                w.writerow([r[k] for k in keep])
summ = subprocess.run([sys.executable, os.path.join(os.path.dirname(__file__), "pmc_summary.py"), src],
                      capture_output=True, text=True, check=True).stdout
json.loads(summ)
open(os.path.join(dst, prefix + "_pmc_summary.json"), "w").write(summ)
print("wrote", sorted(os.listdir(dst)))
