import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")
# Which network kernel a test exercises is the test's choice, not the model load's: loads do not compile / fetch a graph's own
# kernel by themselves here (the library's default does, for every graph but kws_conv); tests/test_gpu_net_jit.py asks for it.
os.environ.setdefault("EDISON_NET_SPECIALIZE", "0")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def mfcc_golden():
    return np.load(os.path.join(GOLDEN, "mfcc_golden.npz"))


@pytest.fixture(scope="session")
def cnn_golden():
    return np.load(os.path.join(GOLDEN, "cnn_golden.npz"))


@pytest.fixture(scope="session")
def kws_golden():
    return np.load(os.path.join(GOLDEN, "kws_golden.npz"))


@pytest.fixture(scope="session")
def q15_golden():
    """Variant C (firmware Q15 MFCC): tests/golden/gen_fixtures_q15.py"""
    return np.load(os.path.join(GOLDEN, "mfccq15_golden.npz"))


@pytest.fixture(scope="session")
def oracle_mod():
    from oracle import oracle
    oracle.build()
    return oracle


@pytest.fixture(scope="session")
def oracle_model(oracle_mod):
    return oracle_mod.Model()


@pytest.fixture(scope="session")
def built_lib():
    """The C-ABI shared library (built with hipcc if the tree does not hold it yet)."""
    # torch ships its own copy of the HIP runtime; a process that also uses torch tensors must load torch's copy
    # FIRST, or torch later finds "No HIP GPUs" (two runtimes cannot both own the device). The library itself
    # never needs torch.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    from edison_amd import build as edbuild, _lib
    edbuild.build()
    return _lib.lib()


@pytest.fixture(scope="session")
def ctx(built_lib):
    """A libedison_hip context on GPU 0 (gpu tests only)."""
    from edison_amd.context import default_context
    return default_context()
