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
    from oracle import pyoracle
    pyoracle.lib()
    return pyoracle


@pytest.fixture(params=["served", "lockstep"])
def multi_mode(request, monkeypatch):
    """How cook_cycle_match_multi places several pools: served walkers (one persistent walker workgroup per pool beside serve launches;
    the default) or lockstep launches (COOK_MATCH_SERVED=0, also what finishes a served match that gave up).  Same results either way."""
    monkeypatch.setenv("COOK_MATCH_SERVED", "1" if request.param == "served" else "0")
    return request.param
