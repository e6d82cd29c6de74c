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
    """CPU-side artefacts every test may need: host tools and the C oracle (seconds to build)."""
    subprocess.run(["make", "-s", "-C", os.path.join(REPO, "bwa-meme_amd"), "host"], check=True)
    subprocess.run(["make", "-s", "-f", "oracle/Makefile"], cwd=REPO, check=True)
