import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session", autouse=True)
def _built_artifacts():
    """(Re)build the HIP library and the oracle when their sources are newer than the binaries, so the
    tests never run against a stale libnsc_hip.so.  hipcc cross-compiles without a GPU."""
    import shutil
    from neural_spectral_codec_amd import build
    if build.is_stale() and (shutil.which("hipcc") or os.path.exists("/opt/rocm/bin/hipcc")):
        build.build_hip()
    import nsc_oracle
    nsc_oracle.build()
    yield


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


def has_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False
