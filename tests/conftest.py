import os
import sys

import pytest

# torch ships its own copy of the HIP runtime; if libstn.so pulls in the system one first, a later `import torch` in the same
# process finds no GPU.  A few GPU tests use torch tensors as gather payloads, so torch goes first in every session (bench.py
# does the same).
try:
    import torch  # noqa: F401
except ImportError:  # CPU-only environments without torch still run the host tests
    pass

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden():
    import json
    with open(os.path.join(ROOT, "tests", "golden", "host_fixtures.json"), encoding="utf-8") as f:
        return json.load(f)
