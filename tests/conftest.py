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
    """Oracle (checker) and the HIP library are built in-tree.  `make` runs every session (it is incremental: a no-op
    when the shared objects are newer than their sources, which is the case on the GPU box where the prebuilt files
    travel with the snapshot), so an edited source can never be tested against a stale .so."""
    subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "oracle")], check=True)
    subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "flagger_amd", "csrc")], check=True)
    yield
