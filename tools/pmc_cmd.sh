#!/bin/bash
# Kernel stats + two PMC passes of ANY python command of this repository, on the GPU box (separate passes, never
# combined with trace domains other than --kernel-trace; the program itself after `--`).  Outputs under
# gpurun_out/prof/<tag>/; tools/pmc_summary.py <dir> --kernel <substring> summarises one kernel of it.
# Usage: tools/pmc_cmd.sh <tag> <script.py> [args...]     e.g. tools/pmc_cmd.sh r05_batch tools/batch_fits.py 300 40 wave
set -uo pipefail
TAG="$1"; shift
REPO="${GRAFT_REPO_ROOT:-$(pwd)}"
OUT="$REPO/gpurun_out/prof/$TAG"
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
SCRIPT="$REPO/$1"; shift
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- python3 "$SCRIPT" "$@" > "$OUT/trace.log" 2>&1
echo "trace rc=$?"
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAVES --output-format csv -d "$OUT/pmc_sq" -- python3 "$SCRIPT" "$@" > "$OUT/pmc_sq.log" 2>&1
echo "pmc_sq rc=$?"
rocprofv3 --kernel-trace --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM --output-format csv -d "$OUT/pmc_lds" -- python3 "$SCRIPT" "$@" > "$OUT/pmc_lds.log" 2>&1
echo "pmc_lds rc=$?"
