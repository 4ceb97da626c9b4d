import importlib
import os
import sys

import pytest
import torch  # noqa: F401  -- before libcrsdr.so: torch bundles its own HIP runtime under the same soname, and its
#                             device init fails when the system runtime (which libcrsdr.so would pull in) is loaded first

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def synth():
    return importlib.import_module("coherent-rtlsdr_amd.synth")


@pytest.fixture(scope="session")
def oracle():
    import oracle_py
    oracle_py.build()
    return oracle_py


@pytest.fixture(scope="session")
def model():
    import model_fp64
    return model_fp64


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")
