import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
DIAG = os.path.join(ROOT, "tools", "diag")       # diagnostics two tests reuse (the fuzzer's case generator, the kernels' resource table)
if DIAG not in sys.path:
    sys.path.append(DIAG)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    """The CPU oracle (test infrastructure): built on demand with gcc."""
    from oracle import oracle as O
    O.lib()
    return O


@pytest.fixture(scope="session")
def engine():
    """The product library; on a GPU box it must exist and a device must be visible."""
    import icp_amd
    icp_amd.lib()
    return icp_amd
