"""CPU: the C-ABI libraries load and export every symbol their headers declare (no compute calls)."""
import os
import re

from sbwt_amd import capi, hostlib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared(header, prefix):
    text = open(os.path.join(ROOT, "include", header)).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(%s\w+)\s*\(" % prefix, text)))


def test_sbwtgpu_exports_every_declared_symbol():
    L = capi.lib()
    names = declared("sbwtgpu.h", "sbwtgpu_")
    assert len(names) >= 20
    for n in names:
        assert hasattr(L, n), n
    assert sorted(capi.EXPORTED_SYMBOLS) == names
    assert b"gfx950" in L.sbwtgpu_version()


def test_sbwthost_exports_every_declared_symbol():
    H = hostlib.lib()
    names = declared("sbwthost.h", "sbwthost_")
    for n in names:
        assert hasattr(H, n), n
    assert sorted(hostlib.EXPORTED_SYMBOLS) == names


def test_no_gpu_is_an_error_not_a_fallback():
    # without a device index_create must fail with NO_DEVICE; with one this test is a no-op
    import numpy as np
    if capi.device_count() > 0:
        return
    w = np.array([1], dtype=np.uint64)
    try:
        capi.Index.create(w, w, w, w, None, 10, 3)
    except capi.SbwtGpuError as e:
        assert e.code == capi.ERR_NO_DEVICE
    else:
        raise AssertionError("index_create succeeded without a GPU")


def test_product_does_not_reference_the_oracle():
    # the oracle is test infrastructure: nothing under sbwt_amd/ may include, import or link it
    bad = []
    for d, _, files in os.walk(os.path.join(ROOT, "sbwt_amd")):
        for f in files:
            if f == "build.py":       # the build driver compiles the checker; building is not using
                continue
            if f.endswith((".py", ".hh", ".h", ".cpp", ".hip")):
                t = open(os.path.join(d, f), errors="replace").read()
                if re.search(r"sbwt_oracle|liboracle|from oracle|import oracle|orc_", t):
                    bad.append(os.path.join(d, f))
    assert not bad, bad


def test_search_workspace_is_monotone_in_the_batch_size():
    # pure arithmetic, no GPU: a workspace sized for the largest batch must serve every smaller one -- the packed bases,
    # the list of handed-on reads and the piece table of long reads (whose zone length changes with the batch size) all
    # grow with total_bases
    from sbwt_amd import capi
    sizes = sorted(set([0, 1, 31, 32, 1000, 1 << 20, (1 << 25) - 1, 1 << 25, (1 << 25) + 1, 1 << 26, 1 << 27, (1 << 28) + 5,
                        1 << 30, 3 << 30, (1 << 36) - 1] + [int(1.37 ** e) for e in range(5, 78)]))
    prev = -1
    for b in sizes:
        w = capi.search_workspace_bytes(b)
        assert w >= prev, (b, w, prev)
        assert w % 256 == 0 and w >= b // 2
        prev = w
    assert capi.search_workspace_bytes(1_500_000_000) < 1_500_000_000      # 0.66 bytes per base + a constant


def test_replication_plan_dedups_devices_and_finds_the_root():
    """sbwtgpu_index_bcast (the single-process RCCL replication of the CLI's --gpus N) sends the image once per DISTINCT device:
    the ranks of its RCCL group are the distinct devices in order of first appearance, duplicates of a device share its handle,
    the root's rank is its device's place among them.  The plan is a pure function (sbwtgpu_debug_bcast_plan): tested here
    without any GPU; the RCCL calls themselves need two devices (tests/test_gpu_multi.py)."""
    import ctypes as C
    from sbwt_amd import capi
    L = capi.lib()
    fn = L.sbwtgpu_debug_bcast_plan
    fn.restype = C.c_int

    def plan(devs, root, visible):
        n = len(devs)
        a = (C.c_int * n)(*devs)
        uniq, slot = (C.c_int * n)(), (C.c_int * n)()
        nu, rr = C.c_int(-5), C.c_int(-5)
        rc = fn(n, a, root, visible, uniq, slot, C.byref(nu), C.byref(rr))
        return rc, list(uniq)[: max(nu.value, 0)], list(slot), rr.value

    assert plan([0, 1, 2, 3], 0, 8) == (0, [0, 1, 2, 3], [0, 1, 2, 3], 0)
    assert plan([0, 0], 0, 1) == (0, [0], [0, 0], 0)                                   # two host threads on one GPU: no RCCL at all
    assert plan([2, 0, 2, 1, 0, 2], 1, 4) == (0, [2, 0, 1], [0, 1, 0, 2, 1, 0], 2)     # first appearance orders the ranks
    assert plan([3, 3, 1], 3, 4) == (0, [3, 1], [0, 0, 1], 0)
    rc, _, _, _ = plan([0, 1], 2, 4)                                                   # the root's device must be listed
    assert rc != 0 and b"root" in L.sbwtgpu_last_error()
    rc, _, _, _ = plan([0, 4], 0, 4)                                                   # a device that is not there
    assert rc != 0 and b"out of range" in L.sbwtgpu_last_error()
    rc, _, _, _ = plan([-1], 0, 4)
    assert rc != 0
