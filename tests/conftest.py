import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


@pytest.fixture(scope='session')
def built_lib():
    """Build (if stale) and load libvcgan_hip.so; the CPU suite only checks loading/symbols."""
    import __graft_entry__ as ge
    ge.build()
    import vcgan_pkg  # noqa: F401
    from voice_converter_cyclegan_amd import _lib
    return _lib
