"""GPU fuzz under the driver's -m gpu suite (tools/fuzz_gpu.py, bounded): random index shapes (k 4..63, single / related /
star genomes, tandem repeats, many short sequences, with and without streaming support and reverse complements), random
builder knobs (path_lookahead, path_safe, image_level, path_stitch) and read shapes (fixed and ragged lengths,
substitutions, N, lower case, long reads cut into pieces); every route (5 fused, 4 two-pass path kernel, 1 blocks) must
equal the reference-order kernel bit for bit, and a sample the oracle.  Three fixed seeds x ~20 s: the cases are
reproducible (SEED=n python tools/fuzz_gpu.py)."""
import os
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BUDGET_S = float(os.environ.get("SBWT_FUZZ_SECONDS", 20))


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_fuzz_all_routes_agree(gpu, seed):
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import fuzz_gpu
    n = fuzz_gpu.fuzz(BUDGET_S, seed)
    assert n >= 5, "the fuzzer got through %d cases only" % n
