import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session", autouse=True)
def _native_code_is_built():
    """The shared libraries and the CLI are built in-tree (they are git-ignored); make sure they are
    there and current before any test loads them.  A no-op when __graft_entry__.build() already ran."""
    from sbwt_amd import build
    build.build_all(force=False)


def _gpu_available() -> bool:
    try:
        from sbwt_amd import capi
        return capi.device_count() > 0
    except Exception:
        return False


@pytest.fixture(scope="session")
def gpu():
    """GPU tests must run the HIP path or fail loudly -- never skip silently on a GPU box."""
    from sbwt_amd import capi
    capi.lib()  # raises ImportError if libsbwtgpu.so is missing
    n = capi.device_count()
    if n <= 0:
        pytest.fail("no HIP device visible: -m gpu tests need a GPU (there is no CPU fallback)")
    # every search first poisons its result range: a result the kernel forgets to write must not pass
    # because an earlier launch left the right value in recycled device memory
    capi.set_tuning("poison_results", 1)
    os.environ["SBWTGPU_POISON_RESULTS"] = "1"      # the CLI subprocesses too
    return 0
