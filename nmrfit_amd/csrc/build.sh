#!/bin/bash
# Builds libnmrfit_amd.so for gfx950 (cross-compiles without a GPU).
# Usage: nmrfit_amd/csrc/build.sh [extra hipcc flags]
#        NMRFIT_LIBNAME=libab_x.so nmrfit_amd/csrc/build.sh -DNMRFIT_INTERLEAVE=2   (A/B builds for tools/ab.py)
set -euo pipefail
HERE="$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)"
ROOT="$(cd "$HERE/../.." && pwd)"
OUT="$ROOT/nmrfit_amd/lib"
mkdir -p "$OUT"
HIPCC="${HIPCC:-/opt/rocm/bin/hipcc}"
"$HIPCC" --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared \
    -ffp-contract=on -fno-fast-math \
    -I"$ROOT/include" -I"$HERE" \
    "$HERE/objective.hip" "$HERE/pso.hip" "$HERE/cabi.hip" "$HERE/comm.hip" -ldl \
    -o "$OUT/${NMRFIT_LIBNAME:-libnmrfit_amd.so}" "$@"
echo "built $OUT/${NMRFIT_LIBNAME:-libnmrfit_amd.so}"
