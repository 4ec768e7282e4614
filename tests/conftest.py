import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle as O
    O.build()
    return O


@pytest.fixture(scope="session")
def spm():
    # torch ships a HIP runtime of its own; libspm_hip.so links the system one.  The two coexist when torch initialises
    # first (as in bench.py), so a test process that uses both lets it: tests that allocate torch tensors on the GPU must not
    # depend on another test having touched torch before them.
    try:
        import torch
        torch.cuda.is_available() and torch.cuda.current_device()
    except Exception:
        pass
    import libspm_amd as S
    S.capi.lib()  # fails loudly if libspm_hip.so is missing
    return S


@pytest.fixture(scope="session")
def ctx(spm):
    c = spm.Context(0)
    yield c
    c.close()
