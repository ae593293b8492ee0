#!/bin/bash
# A/B sweep on the GPU box: kernel variants x launch geometry (NMRFIT_TARGET_WAVES).
# Prints kernel_ms and units/s per configuration (bench.py, CPU baseline off).
set -uo pipefail
REPO="${GRAFT_REPO_ROOT:-$(pwd)}"
cd "$REPO"
for variant in ${VARIANTS:-0 3}; do
  for tw in ${TARGETS:-0 4096 8192 16384 32768}; do
    if [ "$tw" = "0" ]; then unset NMRFIT_TARGET_WAVES; else export NMRFIT_TARGET_WAVES=$tw; fi
    python3 bench.py --steps ${STEPS:-10} --warmup 2 --cpu-seconds 0 --variant $variant ${BENCH_ARGS:-} 2>&1 | python3 -c "
import sys, json
for line in sys.stdin:
    line=line.strip()
    if line.startswith('{'):
        d=json.loads(line)
        r=d['roofline']
        print('variant $variant target_waves $tw: kernel_ms %.4f  gen_ms %.4f  units/s %.4g  hbm_frac %.3f  waves %d nseg %d' % (r['kernel_ms'], d['ms_per_step'], d['value'], r['frac'], r['launch']['waves'], r['launch']['segments']))
    elif line: print(line)
"
  done
done
