import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "slow: full-size runs (a minute or two each on the GPU box); still part of -m gpu")


@pytest.fixture(scope="session")
def built_libs():
    """Build what is missing (HIP library cross-compiles without a GPU; the oracle needs gcc only)."""
    import subprocess
    lib = os.path.join(ROOT, "simlod_amd", "lib", "libsimlod_hip.so")
    if not os.path.exists(lib):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "simlod_amd", "csrc")])
    port = os.path.join(ROOT, "oracle", "libsimlod_oracle.so")
    if not os.path.exists(port):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "port"])
    return True
