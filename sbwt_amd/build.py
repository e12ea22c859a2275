"""Build driver: compiles every native piece IN-TREE (the .so files travel to the GPU box with the
snapshot; they are git-ignored).

  sbwt_amd/lib/libsbwtgpu.so   HIP kernels + C ABI (include/sbwtgpu.h), hipcc --offload-arch=gfx950
  sbwt_amd/lib/libsbwthost.so  GPU-free host helpers (include/sbwthost.h), g++
  sbwt_amd/bin/sbwt            the `sbwt search|build` CLI (C++ host mirror), g++ linked to libsbwtgpu.so
  oracle/liboracle.so          the CPU oracle (test infrastructure), gcc

Usage: python -m sbwt_amd.build [--force]
"""
from __future__ import annotations

import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "sbwt_amd", "csrc")
HOST = os.path.join(CSRC, "host")
LIB = os.path.join(ROOT, "sbwt_amd", "lib")
BIN = os.path.join(ROOT, "sbwt_amd", "bin")
INC = os.path.join(ROOT, "include")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
CXX = os.environ.get("CXX", "g++")


def _newer(target: str, deps) -> bool:
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def _run(cmd):
    print("+", " ".join(cmd), flush=True)
    subprocess.check_call(cmd)


def _glob(d, exts):
    return [os.path.join(d, f) for f in sorted(os.listdir(d)) if f.endswith(exts)]


def build_gpu(force=False) -> str:
    os.makedirs(LIB, exist_ok=True)
    out = os.path.join(LIB, "libsbwtgpu.so")
    srcs = [os.path.join(CSRC, f) for f in ("sbwt_search.hip", "sbwt_search_fused.hip", "sbwt_api_kernels.hip", "sbwt_derived.hip", "sbwt_build.hip", "sbwt_sort.hip",
                                            "sbwt_format.hip", "sbwtgpu_capi.cpp")]
    deps = srcs + [os.path.join(CSRC, f) for f in ("sbwt_device.h", "sbwt_kernels_common.h", "sbwt_scan.h", "sbwt_search_fused_loop.inc")] + \
        [os.path.join(INC, "sbwtgpu.h")]
    if force or _newer(out, deps):
        _run([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-o", out] + srcs + ["-ldl"])
    return out


def build_host(force=False) -> str:
    os.makedirs(LIB, exist_ok=True)
    out = os.path.join(LIB, "libsbwthost.so")
    src = os.path.join(HOST, "host_capi.cpp")
    deps = [src, os.path.join(INC, "sbwthost.h")] + _glob(HOST, (".hh",))
    if force or _newer(out, deps):
        _run([CXX, "-O3", "-std=c++17", "-fPIC", "-shared", "-pthread", "-Wall", "-o", out, src, "-lz"])
    return out


def build_cli(force=False) -> str:
    os.makedirs(BIN, exist_ok=True)
    out = os.path.join(BIN, "sbwt")
    src = os.path.join(HOST, "sbwt_cli.cpp")
    deps = [src, os.path.join(INC, "sbwtgpu.h"), os.path.join(LIB, "libsbwtgpu.so")] + _glob(HOST, (".hh",))
    if force or _newer(out, deps):
        _run([CXX, "-O3", "-std=c++17", "-pthread", "-Wall", "-o", out, src, "-L" + LIB, "-lsbwtgpu", "-lz",
              "-Wl,-rpath,$ORIGIN/../lib", "-Wl,-rpath,/opt/rocm/lib"])
    return out


def build_oracle(force=False) -> str:
    d = os.path.join(ROOT, "oracle")
    if force:
        subprocess.call(["make", "-C", d, "clean"], stdout=subprocess.DEVNULL)
    _run(["make", "-C", d, "liboracle.so"])
    return os.path.join(d, "liboracle.so")


def build_all(force=False):
    return [build_gpu(force), build_host(force), build_cli(force), build_oracle(force)]


if __name__ == "__main__":
    build_all("--force" in sys.argv)
    print("build ok")
