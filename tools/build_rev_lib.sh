#!/bin/bash
# (container side) builds libsbwtgpu of an earlier commit into sbwt_amd/lib/<name>.so, for same-box A/B runs with
# tools/ab_libs5.sh (the .so travels to the GPU box with the snapshot; it is git-ignored):
#   tools/build_rev_lib.sh <git revision> <name>          e.g.  tools/build_rev_lib.sh 888bb2b lib_r4
set -eu
REV=$1; NAME=$2
ROOT=$(cd "$(dirname "$0")/.." && pwd)
W=$(mktemp -d /tmp/sbwt_rev_XXXX)
git -C "$ROOT" worktree add -f "$W" "$REV" > /dev/null
( cd "$W/sbwt_amd/csrc" && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -o "$ROOT/sbwt_amd/lib/$NAME.so" \
    sbwt_search.hip sbwt_search_fused.hip sbwt_api_kernels.hip sbwt_derived.hip sbwt_build.hip sbwt_sort.hip sbwt_format.hip sbwtgpu_capi.cpp -ldl )
git -C "$ROOT" worktree remove --force "$W"
echo "built $ROOT/sbwt_amd/lib/$NAME.so from $REV"
