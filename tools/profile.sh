#!/bin/bash
# Profiles bench.py on the GPU box: kernel trace + stats, then separate PMC passes
# (never combined with trace domains other than --kernel-trace).  Outputs under gpurun_out/prof/.
# Usage: tools/profile.sh [tag] [bench args...]      e.g. tools/profile.sh r03            (the headline kernel)
#                                                        tools/profile.sh r03ff --variant 6  (the far-field kernel fit() picks at C3 size)
# --no-other-configs everywhere: every launch of the profiled objective kernel then has the C3 shape, so the
# --stats average of that kernel is the number bench.py reports as roofline.kernel_ms.
set -uo pipefail
TAG="${1:-r01}"; shift || true
REPO="${GRAFT_REPO_ROOT:-$(pwd)}"
OUT="$REPO/gpurun_out/prof/$TAG"
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
ARGS="--steps 10 --warmup 2 --cpu-seconds 0 --no-extras --no-other-configs --no-pmc $*"
# kernel trace + stats of the DEFAULT command (what the driver runs); the PMC passes below use a
# shorter form of it (no CPU baseline, no extras)
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- python3 "$REPO/bench.py" --no-other-configs --no-pmc "$@" > "$OUT/trace.log" 2>&1
echo "trace rc=$?"
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAVES --output-format csv -d "$OUT/pmc_sq" -- python3 "$REPO/bench.py" $ARGS > "$OUT/pmc_sq.log" 2>&1
echo "pmc_sq rc=$?"
rocprofv3 --kernel-trace --pmc FETCH_SIZE GRBM_GUI_ACTIVE --output-format csv -d "$OUT/pmc_fetch" -- python3 "$REPO/bench.py" $ARGS > "$OUT/pmc_fetch.log" 2>&1
echo "pmc_fetch rc=$?"
rocprofv3 --kernel-trace --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d "$OUT/pmc_write" -- python3 "$REPO/bench.py" $ARGS > "$OUT/pmc_write.log" 2>&1
echo "pmc_write rc=$?"
rocprofv3 --kernel-trace --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM --output-format csv -d "$OUT/pmc_lds" -- python3 "$REPO/bench.py" $ARGS > "$OUT/pmc_lds.log" 2>&1
echo "pmc_lds rc=$?"
find "$OUT" -name "*.csv" | head -40
