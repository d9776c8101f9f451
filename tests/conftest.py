import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def reflib():
    """The reference itself (oracle/_ref/liblis_ref.so), 1 OpenMP thread.  Skips when not built."""
    import orc
    import lisdrv
    if not os.path.exists(orc.REF_SO):
        try:
            orc.build()
        except Exception:
            pass
    if not os.path.exists(orc.REF_SO):
        pytest.skip("oracle/_ref/liblis_ref.so not built (no /root/reference here)")
    return lisdrv.open_lib(orc.REF_SO, threads=1)
