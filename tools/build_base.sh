#!/bin/bash
# Builds the library of a commit (default HEAD) as raytracing_amd/lib/librt_hip_base.so, for A/B runs
# against the working tree: tools/build_base.sh [commit]
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
REV=${1:-HEAD}
TMP=$(mktemp -d /tmp/rt_base_XXXX)
git -C "$ROOT" archive "$REV" raytracing_amd/csrc include | tar -x -C "$TMP"
make -j8 -C "$TMP/raytracing_amd/csrc" >/dev/null
cp "$TMP/raytracing_amd/lib/librt_hip.so" "$ROOT/raytracing_amd/lib/librt_hip_base.so"
rm -rf "$TMP"
echo "librt_hip_base.so = $REV"
