#!/bin/bash
# SQ_INSTS_VALU of the batched kernel (wave = particle form, 40 fits x 204 particles) for a few (N, P): the counts
# fit   instructions per particle and generation = a + chunks * (b + c * P)   -- a the swarm step's prologue and
# epilogue, b a chunk's per-point work (phase, data, residual), c one peak on one chunk.
# Usage (GPU box): tools/batch_instr_model.sh ; python tools/batch_instr_model.py gpurun_out/prof/instr_model
set -uo pipefail
REPO="${GRAFT_REPO_ROOT:-$(pwd)}"
OUT="$REPO/gpurun_out/prof/instr_model"
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
for cfg in "4096 1" "4096 6" "4096 12" "8192 1" "8192 6" "8192 12" "2048 6"; do
    set -- $cfg
    export BF_N=$1 BF_P=$2
    rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_WAVES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU --output-format csv -d "$OUT/n$1_p$2" -- python3 "$REPO/tools/batch_fits.py" 100 40 wave > "$OUT/n$1_p$2.log" 2>&1 || exit 1
    echo "N=$1 P=$2 done"
done
