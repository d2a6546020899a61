import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session", autouse=True)
def _built():
    """Oracle (checker) and the HIP library are built in-tree; on the GPU box the prebuilt .so files
    travel with the snapshot, so this is a no-op there."""
    if not os.path.exists(os.path.join(ROOT, "oracle", "liboracle_hf.so")):
        subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "oracle")], check=True)
    if not os.path.exists(os.path.join(ROOT, "flagger_amd", "csrc", "libhmmflagger_hip.so")):
        subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "flagger_amd", "csrc")], check=True)
    yield
