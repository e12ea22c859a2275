#!/bin/bash
# libsbwtgpu built with extra -D switches, for A/B runs: tools/build_variant_lib.sh <name> [-DX=1 ...]  ->  sbwt_amd/lib/lib_<name>.so
NAME=$1; shift
cd "$(dirname "$0")/.." && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -w "$@" -o sbwt_amd/lib/lib_$NAME.so \
  sbwt_amd/csrc/sbwt_search.hip sbwt_amd/csrc/sbwt_search_fused.hip sbwt_amd/csrc/sbwt_api_kernels.hip sbwt_amd/csrc/sbwt_derived.hip \
  sbwt_amd/csrc/sbwt_build.hip sbwt_amd/csrc/sbwt_sort.hip sbwt_amd/csrc/sbwt_format.hip sbwt_amd/csrc/sbwtgpu_capi.cpp -ldl
