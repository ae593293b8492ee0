#!/bin/bash
# Kernel durations of a small-swarm run (rocprofv3 --kernel-trace --stats): how much of a generation is the kernel itself.
#   bash tools/small_swarm_kernel_trace.sh [S N P]        (on the GPU box; output under gpurun_out/)
set -uo pipefail
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
S="${1:-204}"; N="${2:-4096}"; P="${3:-6}"
cat > /tmp/ss_one.py <<PY
import os, sys, time
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
from nmrfit_amd import synth, pso
from nmrfit_amd.equations import Evaluator
sp = synth.make_spectrum($N, $P, seed=1)
with Evaluator(sp["w"], sp["u"], sp["v"], sp["weights"]) as ev:
    sw = pso.DeviceSwarm(ev, sp["lower"], sp["upper"], $S, seed=3, minfunc=-1.0, minstep=-1.0)
    sw.run(50, check_every=50)
    t0 = time.perf_counter(); sw.run(2000, check_every=100); dt = time.perf_counter() - t0
    print("S=$S N=$N P=$P: %.2f us per generation, launches per generation %d" % (dt / 2000 * 1e6, sw.last_launches()))
    sw.close()
PY
export TMPDIR=/tmp; rm -rf /tmp/sstrace
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/sstrace -- python3 /tmp/ss_one.py) 2>&1 | grep "per generation"
python3 - <<'PY'
import csv, glob
f = glob.glob("/tmp/sstrace/**/*_kernel_stats.csv", recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:5]:
    print("%-70s calls %6s avg %8.2f us min %8.2f max %8.2f" % (r["Name"][:70], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3))
PY
