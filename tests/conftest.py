import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session", autouse=True)
def _oracle_threads():
    """The CPU oracle (torch conv3d) is SLOWER with 128-256 threads than with 8-16 on the problem sizes of this suite (GPU box,
    EPYC 9575F: 3.9 s per 128^3 model-A forward on all cores, 2.1 s with 8 threads; a 64^3 forward 1.8 s against 0.3 s): cap the
    intra-op pool so that the oracle forwards of the GPU tests take a fifth of the time.  No effect on results."""
    try:
        import torch
        if torch.get_num_threads() > 16:
            torch.set_num_threads(16)
    except Exception:
        pass
    yield


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
