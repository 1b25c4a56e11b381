#!/bin/sh
# oracle/_ref: the part of the reference that compiles from its own source without R / Rcpp / Eigen -- the `rng`
# class of src/singlet.cpp (xorshift hash behind the cross-validation mask and the synthetic generator).
# The class text is cut out of the reference tree AT BUILD TIME into the git-ignored oracle/_ref/ and compiled with
# oracle/ref_rng_shim.cpp; no reference source is copied into the repository.  Needs /root/reference (the authoring
# container); on the GPU box the prebuilt oracle/_ref/librng_ref.so travels with the snapshot.
# The ALS functions of src/singlet.cpp need RcppEigen + R and are NOT buildable here (DESIGN.md "Oracle status").
set -e
HERE=$(cd "$(dirname "$0")" && pwd)
REF=${SINGLET_REFERENCE:-/root/reference}
SRC="$REF/src/singlet.cpp"
if [ ! -f "$SRC" ]; then
    echo "make_ref.sh: $SRC not present (not the authoring container): keeping any prebuilt oracle/_ref" >&2
    exit 0
fi
mkdir -p "$HERE/_ref"
# from the line `class rng {` to the first line that is exactly `};`
awk '/^class rng \{/ {on = 1} on {print} on && /^\};/ {exit}' "$SRC" > "$HERE/_ref/rng_class.inc"
grep -q "uint64_t rand(uint64_t i, uint64_t j)" "$HERE/_ref/rng_class.inc" || { echo "make_ref.sh: rng class not found in $SRC" >&2; exit 1; }
${CXX:-g++} -O2 -std=c++17 -fPIC -shared -fvisibility=hidden -w -I"$HERE" -o "$HERE/_ref/librng_ref.so" "$HERE/ref_rng_shim.cpp"
echo "built $HERE/_ref/librng_ref.so from $SRC"
