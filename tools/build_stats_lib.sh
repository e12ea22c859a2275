#!/bin/bash
# libsbwtgpu with the lane-iteration counters compiled in (-DSBWT_STATS): sbwt_amd/lib/lib_stats.so;
# `tools/build_stats_lib.sh timeline`: with the waves' start / drained / exit clocks instead (-DSBWT_TIMELINE): lib_timeline.so
DEF=-DSBWT_STATS; OUT=lib_stats.so
if [ "$1" = timeline ]; then DEF=-DSBWT_TIMELINE; OUT=lib_timeline.so; fi
cd "$(dirname "$0")/.." && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared $DEF -o sbwt_amd/lib/$OUT \
  sbwt_amd/csrc/sbwt_search.hip sbwt_amd/csrc/sbwt_search_fused.hip sbwt_amd/csrc/sbwt_api_kernels.hip sbwt_amd/csrc/sbwt_derived.hip \
  sbwt_amd/csrc/sbwt_build.hip sbwt_amd/csrc/sbwt_sort.hip sbwt_amd/csrc/sbwt_format.hip sbwt_amd/csrc/sbwtgpu_capi.cpp -ldl
