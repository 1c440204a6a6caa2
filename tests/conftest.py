from pathlib import Path
import sys

import pytest

ROOT = Path(__file__).resolve().parents[1]
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with gpurun)")


@pytest.fixture(scope="session")
def oracle():
    """The CPU checkers (test infrastructure, see oracle/lbl_oracle.c)."""
    from oracle import oracle as module
    module.port_library()
    return module
