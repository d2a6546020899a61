import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


# hf_create computes every window's packed record on the host with the function k_setup runs on the device; under this switch it
# also downloads the device's records and refuses to continue if a single bit differs (csrc/hf_estep.hip window_record)
os.environ.setdefault("HF_CREATE_VERIFY", "1")


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
