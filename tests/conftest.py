import os
import sys

import pytest

try:
    # before libmgx.so is ever loaded: the torch wheel and libmgx bring a HIP runtime each, the one
    # loaded first serves both, and torch's device layer only comes up on its own
    import torch  # noqa: F401
except Exception:  # pragma: no cover - torch is optional for the single-GPU tests
    torch = None

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def pkg():
    """The product package (ctypes view of libmgx.so)."""
    import __graft_entry__ as ge

    if not os.path.exists(os.path.join(ge.PKG_DIR, "libmgx.so")):
        ge.build()
    return ge.load_package()


@pytest.fixture(scope="session")
def po():
    """The CPU oracle (test infrastructure; parity unpinned, see oracle/mg_oracle.h)."""
    from oracle import pyoracle

    pyoracle.build()
    return pyoracle


def have_gpu() -> bool:
    try:
        import torch

        return torch.cuda.is_available()
    except Exception:
        return False
