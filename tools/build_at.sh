#!/bin/bash
# Build libnmrfit_amd of an earlier commit under another name, for A/B runs against the working tree:
#   tools/build_at.sh <commit> [hipcc flags]   ->  nmrfit_amd/lib/libab_<commit>.so
set -euo pipefail
ROOT="$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)"
C="$1"; shift
T="$(mktemp -d)"
mkdir -p "$T/csrc" "$T/include"
for f in $(git -C "$ROOT" ls-tree --name-only "$C" nmrfit_amd/csrc/); do git -C "$ROOT" show "$C:$f" > "$T/csrc/$(basename "$f")"; done
for h in $(git -C "$ROOT" ls-tree --name-only "$C" include/); do git -C "$ROOT" show "$C:$h" > "$T/include/$(basename "$h")"; done
# (every translation unit the commit had: one objective.hip up to round 4, one per kernel variant since round 5)
SRCS=$(ls "$T"/csrc/*.hip | grep -v objective_ab.hip)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -ffp-contract=on -fno-fast-math \
    -I"$T/include" -I"$T/csrc" $SRCS -ldl -o "$ROOT/nmrfit_amd/lib/libab_$C.so" "$@"
rm -rf "$T"
echo "built nmrfit_amd/lib/libab_$C.so"
