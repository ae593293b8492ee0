#!/bin/bash
# Counters of several builds of the library in ONE rocprofv3 pass (tools/ab_pmc.py).  Usage on the GPU box:
#   bash tools/run_pmc_ab.sh <variant> lib1.so lib2.so ...      -> gpurun_out/pmc_ab.txt
set -uo pipefail
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
V="$1"; shift
LIBS=""; for l in "$@"; do LIBS="$LIBS $GRAFT_REPO_ROOT/$l"; done
export TMPDIR=/tmp
rm -rf /tmp/abp
(cd /tmp && rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY --output-format csv -d /tmp/abp -- python3 "$GRAFT_REPO_ROOT/tools/ab_pmc.py" --variant "$V" $LIBS > "$GRAFT_REPO_ROOT/gpurun_out/pmc_ab.log" 2>&1)
python3 tools/ab_pmc.py --summarise /tmp/abp "$@" | tee gpurun_out/pmc_ab.txt
