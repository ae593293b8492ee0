#!/bin/bash
# Kernel trace of the device-resident loop for small swarms: device-busy time vs wall per generation.
set -uo pipefail
REPO="${GRAFT_REPO_ROOT:-$(pwd)}"
OUT="$REPO/gpurun_out/prof/small"
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- python3 "$REPO/tools/small_swarm_timing.py" > "$OUT/trace.log" 2>&1
cat "$OUT/trace.log" | tail -6
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + "/trace/*/*_kernel_trace.csv")[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# split into runs by large gaps; report per-kernel mean duration and mean gap between consecutive kernels
byname = collections.defaultdict(list)
gaps = []
for a, b in zip(rows, rows[1:]):
    g = int(b["Start_Timestamp"]) - int(a["End_Timestamp"])
    if g < 50000:
        gaps.append(g)
for r in rows:
    byname[(r["Kernel_Name"].split("(")[0][-40:], r["Grid_Size_X"] if "Grid_Size_X" in r else r.get("Grid_Size", ""))].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
for k, v in sorted(byname.items(), key=lambda kv: -len(kv[1]))[:14]:
    print("%-42s grid %-8s n=%6d mean %.2f us" % (k[0], k[1], len(v), sum(v) / len(v) / 1e3))
print("mean gap between consecutive kernels: %.2f us (n=%d)" % (sum(gaps) / len(gaps) / 1e3, len(gaps)))
PY
