#!/bin/bash
# Builds libnmrfit_amd.so for gfx950 (cross-compiles without a GPU).  The translation units compile in parallel
# (the objective kernel's instantiations are one unit per selectable variant) and are then linked.
# Usage: nmrfit_amd/csrc/build.sh [extra hipcc flags]
#        nmrfit_amd/csrc/build.sh --ab      the A/B library libnmrfit_amd_ab.so: the product + the A/B kernel variants
#                                           (BASELINE, NOSKIP, SINGLE, QUAD, STAGED) and the diagnostic entry points of
#                                           include/nmrfit_amd_diag.h -- what tools/ab.py, bench.py's `variants` entry
#                                           and the parity tests' reference kernels use
#        NMRFIT_LIBNAME=libab_x.so nmrfit_amd/csrc/build.sh -DSOMETHING=2   (one-off A/B builds for tools/ab.py)
set -euo pipefail
HERE="$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)"
ROOT="$(cd "$HERE/../.." && pwd)"
OUT="$ROOT/nmrfit_amd/lib"
mkdir -p "$OUT"
HIPCC="${HIPCC:-/opt/rocm/bin/hipcc}"
EXTRA=()
LIBNAME="${NMRFIT_LIBNAME:-libnmrfit_amd.so}"
UNITS=(objective objective_default objective_farfield objective_farfield32 objective_norec objective_batch objective_batch_im objective_batch_im2 objective_batch_im2f pso batch result cabi comm)
for arg in "$@"; do
    if [ "$arg" = "--ab" ]; then
        EXTRA+=(-DNMRFIT_AB_BUILD)
        LIBNAME="${NMRFIT_LIBNAME:-libnmrfit_amd_ab.so}"
        UNITS+=(objective_ab)
    else
        EXTRA+=("$arg")
    fi
done
OBJ="$(mktemp -d "${TMPDIR:-/tmp}/nmrfit_build.XXXXXX")"
trap 'rm -rf "$OBJ"' EXIT
# (-ffile-prefix-map: __FILE__ in the error messages is relative to the repository, so the library's bytes do not depend
# on where the repository lies -- it is built here and checked against a rebuild on the GPU box)
FLAGS=(--offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden -ffp-contract=on -fno-fast-math -ffile-prefix-map="$ROOT"=. -I"$ROOT/include" -I"$HERE")
pids=()
for u in "${UNITS[@]}"; do
    [ -f "$HERE/$u.hip" ] || continue
    # (-cuid: the compilation-unit id hipcc otherwise hashes from the command line, temporary directory included; fixed, the
    # same sources give the same device code bytes -- what __graft_entry__.smoke() compares on the GPU box)
    "$HIPCC" "${FLAGS[@]}" "${EXTRA[@]}" -cuid="nmrfit_$u" -c "$HERE/$u.hip" -o "$OBJ/$u.o" &
    pids+=($!)
done
fail=0
for p in "${pids[@]}"; do wait "$p" || fail=1; done
[ "$fail" = 0 ] || { echo "compilation failed" >&2; exit 1; }
"$HIPCC" --offload-arch=gfx950 -fPIC -shared "$OBJ"/*.o -ldl -o "$OUT/$LIBNAME"
# a stamp written by __graft_entry__.build() describes the libraries it built: it no longer describes this one
case "$LIBNAME" in libnmrfit_amd.so|libnmrfit_amd_ab.so) rm -f "$OUT/BUILD_STAMP.json" ;; esac
echo "built $OUT/$LIBNAME"
