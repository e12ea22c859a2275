"""sbwt_amd -- MI355X-native k-mer search path for plain-matrix SBWT indexes.

The product is the C-ABI shared library `sbwt_amd/lib/libsbwtgpu.so` (include/sbwtgpu.h) built
from hand-written HIP kernels in `sbwt_amd/csrc/`, plus the C++ host mirror of the reference's
`sbwt search` / `SBWT` / `SubsetMatrixRank` interface in `sbwt_amd/csrc/host/`.  The Python in
this package is plumbing for tests and bench.py (ctypes binding, synthetic data, build driver).
"""
__version__ = "0.1.0"
