import os
import sys

import pytest

# No torch here: the torch wheel bundles its own libamdhip64 / libhsa-runtime64 / librccl with the SONAMEs of
# the /opt/rocm ones libmgx is built against, and whichever copy is loaded first serves the whole process.
# The GPU tests must exercise the library on ITS stack, so nothing imports torch before libmgx.so is loaded
# (device buffers: tests/hipmem.py; rank processes: the TCP store of rendezvous.py).  The CPU-only plan tests
# (tests/test_dist_plan_cpu.py) do use torch.distributed / gloo: no GPU is touched there.

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
    """is there a HIP device?  Asked of the runtime libmgx links against (tests/hipmem.py)"""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    try:
        import hipmem

        return hipmem.device_count() > 0
    except Exception:
        return False
