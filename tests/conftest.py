import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def amd():
    """The product package; compiles the HIP library first if it is missing or stale (hipcc cross-compiles
    without a GPU, and the GPU box has the same toolchain)."""
    import importlib
    build = importlib.import_module(
        "automated-brain-mri-analysis-and-report-generation-with-retrieval-augmented-clinical-assistance_amd._build")
    build.build()
    import brats_amd
    return brats_amd


@pytest.fixture(scope="session")
def gpu(amd):
    """The HIP path or nothing: GPU tests fail (not skip) if the extension or the device is missing."""
    import torch
    assert torch.cuda.is_available(), "GPU test collected on a machine without a GPU"
    lib = amd._lib.load()
    assert lib.mi355_device_count() > 0
    return torch.device("cuda:0")
