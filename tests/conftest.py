import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


def pytest_sessionstart(session):
    """Keep libnerf_hip.so in step with its sources: `make` is a no-op when it is current, and a box that received the
    tree without the (git-ignored) library, or with one older than csrc/, gets it built before the first test.
    A build failure is not hidden: the tests that need the library then fail loudly in _native.lib()."""
    import shutil
    import subprocess
    csrc = os.path.join(ROOT, "nerf_meets_mlx_amd", "csrc")
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc) or os.environ.get("NERF_SKIP_AUTOBUILD"):
        return
    try:
        subprocess.run(["make", "-C", csrc, "-j", str(min(8, os.cpu_count() or 1))], check=False,
                       stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=1500)
    except Exception:
        pass
