import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.dirname(os.path.abspath(__file__))):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # the built library is git-ignored: a fresh checkout gets it here (hipcc cross-compiles without a GPU)
    lib = os.path.join(ROOT, "uniformgrid-raytracing_amd", "libugrt.so")
    if not os.path.exists(lib):
        import subprocess

        subprocess.run(["make", "-j", "6", "-C", os.path.join(ROOT, "uniformgrid-raytracing_amd", "csrc")], check=True,
                       capture_output=True)


def _have_gpu():
    try:
        import torch

        return torch.cuda.is_available()
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    if _have_gpu():
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for it in items:
        if "gpu" in it.keywords:
            it.add_marker(skip)


@pytest.fixture(scope="session")
def ugrt():
    """The product package (host bindings over libugrt.so)."""
    import importlib

    return importlib.import_module("uniformgrid-raytracing_amd")


@pytest.fixture(scope="session")
def O():
    """The CPU oracle (checker)."""
    import oracle_lib

    return oracle_lib
