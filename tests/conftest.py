import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (gfx950) device")


@pytest.fixture(scope="session")
def O():
    """The CPU oracle (test infrastructure)."""
    from oracle import oracle
    oracle.lib()
    return oracle


@pytest.fixture(scope="session")
def rsa():
    """The product package (radio-sim_amd/)."""
    import radio_sim_amd
    radio_sim_amd.build_library()      # hipcc cross-compiles without a GPU; a no-op when the library is current
    return radio_sim_amd


@pytest.fixture()
def engine(rsa):
    eng = rsa.Engine(0)
    yield eng
    eng.close()
