#!/bin/bash
# Build the library of an earlier commit for A/B runs with tools/ab_lib.py (same box, interleaved rounds):
#   tools/build_ref_lib.sh <commit> <tag>   ->  open-hummingbird-eval_amd/lib/abl/libhbird_hip_<tag>.so
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
TMP=$(mktemp -d)
git -C "$ROOT" archive "$1" open-hummingbird-eval_amd/csrc include | tar -x -C "$TMP"
make -C "$TMP/open-hummingbird-eval_amd/csrc" -j8 > /dev/null
mkdir -p "$ROOT/open-hummingbird-eval_amd/lib/abl"
cp "$TMP/open-hummingbird-eval_amd/lib/libhbird_hip.so" "$ROOT/open-hummingbird-eval_amd/lib/abl/libhbird_hip_$2.so"
echo "built $ROOT/open-hummingbird-eval_amd/lib/abl/libhbird_hip_$2.so"
