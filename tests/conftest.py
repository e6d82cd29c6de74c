import os
import subprocess
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "tests"))
sys.path.insert(0, os.path.join(REPO, "bwa-meme_amd"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session", autouse=True)
def _built_host_and_oracle():
    """Artefacts every test may need: host tools, the C oracle (seconds to build) and -- where hipcc exists -- the HIP
    library (incremental; cross-compiles for gfx950 without a GPU), so that a fresh checkout tests what it contains."""
    import shutil
    subprocess.run(["make", "-s", "-C", os.path.join(REPO, "bwa-meme_amd"), "host"], check=True)
    if shutil.which("hipcc") or os.path.exists("/opt/rocm/bin/hipcc"):
        subprocess.run(["make", "-s", "-C", os.path.join(REPO, "bwa-meme_amd"), "hip"], check=True)
    subprocess.run(["make", "-s", "-f", "oracle/Makefile"], cwd=REPO, check=True)
