#!/bin/bash
# One SQ counter pass (instruction counts / VALU busy) of bench.py with extra args.
# Usage: tools/profile_sq.sh <tag> [bench args...]
set -uo pipefail
TAG="${1:-sq}"; shift || true
REPO="${GRAFT_REPO_ROOT:-$(pwd)}"
OUT="$REPO/gpurun_out/prof/$TAG"
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- python3 "$REPO/bench.py" --steps 10 --warmup 2 --cpu-seconds 0 --no-extras --no-other-configs --no-pmc "$@" > "$OUT/trace.log" 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAVES --output-format csv -d "$OUT/pmc_sq" -- python3 "$REPO/bench.py" --steps 10 --warmup 2 --cpu-seconds 0 --no-extras --no-other-configs --no-pmc "$@" > "$OUT/pmc_sq.log" 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM --output-format csv -d "$OUT/pmc_lds" -- python3 "$REPO/bench.py" --steps 10 --warmup 2 --cpu-seconds 0 --no-extras --no-other-configs --no-pmc "$@" > "$OUT/pmc_lds.log" 2>&1
python3 "$REPO/tools/pmc_summary.py" "$OUT" | python3 -c "
import json,sys
d=json.load(sys.stdin)
for k,v in d['objective_kernel_pmc'].items(): print('%-24s %.6g' % (k, v['mean']))
for k in d:
    if k not in ('kernel_stats','objective_kernel_pmc'): print(k, d[k])
print(d['kernel_stats'][0])"
