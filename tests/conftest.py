import os
from pathlib import Path
import sys

import pytest

ROOT = Path(__file__).resolve().parents[1]
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))


# The MT-CKD coefficient tables (conversion of the reference's data file, see
# tests/golden/make_mt_ckd.py); /root/reference does not exist on the GPU box.
MT_CKD_TABLES = ROOT / "tests" / "golden" / "mt_ckd_bands.npz"
os.environ.setdefault("PYLBL_MT_CKD", str(MT_CKD_TABLES))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with gpurun)")


@pytest.fixture(scope="session")
def oracle():
    """The CPU checkers (test infrastructure, see oracle/lbl_oracle.c)."""
    from oracle import oracle as module
    module.port_library()
    return module


@pytest.fixture(scope="session")
def continuum_oracle():
    """numpy restatement of the MT-CKD path (test infrastructure, oracle/mt_ckd_oracle.py)
    bound to the coefficient fixture."""
    from oracle import mt_ckd_oracle as module
    tables = module.load_tables(str(MT_CKD_TABLES))
    cache = {}

    def continuum(owner):
        if owner not in cache:
            cache[owner] = module.Continuum(owner, tables)
        return cache[owner]
    module.continuum = continuum
    return module
