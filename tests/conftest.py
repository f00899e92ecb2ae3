import os
import sys

import pytest

# libstn.so is built and shipped against the system ROCm.  PyTorch-ROCm wheels bundle another copy of the HIP runtime, and a process
# that imports torch before the library binds the library to that copy.  The product needs no PyTorch (bench.py --gpus 1 and every
# test here run without it; only the N > 1 bench ranks use torch.distributed), so the test session keeps torch OUT of the process:
# the binding's convenience preload is switched off, and the tests that need torch (gloo process groups, the torch statement of the
# stages) are CPU tests that never create an engine, or run in a subprocess (tests/test_gpu_bench_contract.py).
os.environ.setdefault("STN_NO_TORCH_PRELOAD", "1")

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_ignore_collect(collection_path, config):
    """`-m gpu` (the GPU box): only tests/test_gpu_*.py carry that mark, and collecting the CPU files would import torch (gloo tests,
    the torch statement of the ops) into the process before libstn.so is loaded — which is exactly what the GPU run must not do."""
    if config.getoption("-m", default="") == "gpu":
        name = collection_path.name
        if name.startswith("test_") and name.endswith(".py") and not name.startswith("test_gpu_"):
            return True
    return None


@pytest.fixture(scope="session")
def golden():
    import json
    with open(os.path.join(ROOT, "tests", "golden", "host_fixtures.json"), encoding="utf-8") as f:
        return json.load(f)
