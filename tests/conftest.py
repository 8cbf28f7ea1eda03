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
    A failed build ENDS the session with the compiler output: running the GPU suite against a stale library that no
    longer matches csrc/ would be a false green."""
    import shutil
    import subprocess
    csrc = os.path.join(ROOT, "nerf_meets_mlx_amd", "csrc")
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if os.environ.get("NERF_SKIP_AUTOBUILD"):
        return
    if not os.path.exists(hipcc):
        return      # no compiler on this box: _native.lib() raises if the library is absent
    try:
        r = subprocess.run(["make", "-C", csrc, "-j", str(min(8, os.cpu_count() or 1))], check=False,
                           stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=1500, text=True)
    except subprocess.TimeoutExpired as e:
        pytest.exit(f"building libnerf_hip.so timed out after {e.timeout} s", returncode=3)
    if r.returncode != 0:
        pytest.exit("building libnerf_hip.so failed (make exit %d):\n%s" % (r.returncode, r.stdout[-6000:]), returncode=3)
