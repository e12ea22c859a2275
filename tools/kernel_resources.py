"""Register and LDS use of the search kernels, read from the code objects INSIDE the built library (what the loader runs):
.vgpr_count, .sgpr_count, .sgpr_spill_count, .vgpr_spill_count, LDS (.group_segment_fixed_size) and scratch
(.private_segment_fixed_size) of every kernel whose name matches the pattern, keyed by the full (demangled) name.

    python tools/kernel_resources.py [library.so] [name-regex]        (defaults: sbwt_amd/lib/libsbwtgpu.so, k_search)

tools/summarize_profile.py appends this table to every profiles/*_rocprof_summary.txt, so that the occupancy and spill
claims of DESIGN.md section 3 can be checked from profiles/ (the trace's VGPR_Count row is the granule
of whichever dispatch came last, not the code object's numbers).

The library is a HIP fat binary: one clang offload bundle per translation unit ("__CLANG_OFFLOAD_BUNDLE__", entries of
{ offset, size, triple }); the gfx950 entries are ELF code objects whose NT_AMDGPU_METADATA note carries the numbers
(llvm-readelf --notes prints it as YAML).  Nothing is written next to the library (llvm-objdump --offloading would).
"""
import os
import re
import struct
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
READELF = "/opt/rocm/lib/llvm/bin/llvm-readelf"
CXXFILT = "c++filt"
MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"
FIELDS = (".vgpr_count", ".agpr_count", ".sgpr_count", ".vgpr_spill_count", ".sgpr_spill_count", ".group_segment_fixed_size",
          ".private_segment_fixed_size", ".max_flat_workgroup_size")


def code_objects(path):
    """the gfx950 code objects of every offload bundle in the file"""
    d = open(path, "rb").read()
    out = []
    for m in re.finditer(re.escape(MAGIC), d):
        base = m.start()
        (n,) = struct.unpack_from("<Q", d, base + len(MAGIC))
        p = base + len(MAGIC) + 8
        for _ in range(n):
            off, size, tlen = struct.unpack_from("<QQQ", d, p)
            triple = d[p + 24:p + 24 + tlen].decode()
            p += 24 + tlen
            if "amdgcn" in triple and size:
                out.append(d[base + off:base + off + size])
    return out


def kernels_of(obj_bytes):
    with tempfile.NamedTemporaryFile(suffix=".co") as f:
        f.write(obj_bytes)
        f.flush()
        txt = subprocess.run([READELF, "--notes", f.name], capture_output=True, text=True).stdout
    ks, cur = [], None
    for line in txt.splitlines():
        s = line.strip()
        if s.startswith("- .agpr_count") or s.startswith("- .args"):
            cur = {}
            ks.append(cur)
            s = s[2:]
        if cur is None or ":" not in s:
            continue
        key, val = s.split(":", 1)
        key = key.strip()
        if key in FIELDS or key == ".name":
            cur[key] = val.strip().strip("'")
    return [k for k in ks if ".name" in k]


def demangle(names):
    try:
        out = subprocess.run([CXXFILT], input="\n".join(names), capture_output=True, text=True).stdout.splitlines()
        return out if len(out) == len(names) else names
    except OSError:
        return names


def table(lib, pattern="k_search"):
    rows = []
    for obj in code_objects(lib):
        for k in kernels_of(obj):
            if re.search(pattern, k[".name"]):
                rows.append(k)
    names = demangle([k[".name"] for k in rows])
    lines = []
    for k, nm in sorted(zip(rows, names), key=lambda t: t[1]):
        nm = re.sub(r"^void ", "", re.sub(r"\(.*", "", nm))    # the template arguments say which instantiation it is
        lines.append("%-62s vgpr %3s  sgpr %3s  sgpr_spill %3s  vgpr_spill %3s  lds %6s B  scratch %4s B" % (
            nm[:62], k.get(".vgpr_count", "?"), k.get(".sgpr_count", "?"), k.get(".sgpr_spill_count", "?"),
            k.get(".vgpr_spill_count", "?"), k.get(".group_segment_fixed_size", "?"), k.get(".private_segment_fixed_size", "?")))
    return lines


if __name__ == "__main__":
    lib = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "sbwt_amd", "lib", "libsbwtgpu.so")
    pat = sys.argv[2] if len(sys.argv) > 2 else "k_search"
    print("== code-object resources (%s, kernels matching /%s/) ==" % (os.path.relpath(lib, ROOT) if lib.startswith(ROOT) else lib, pat))
    for ln in table(lib, pat):
        print(ln)
