import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "moss-ttsd_amd")
for p in (ROOT, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")

# The engine reads sealed KV pages only from 512 rows x pages up (below that the passes are latency-bound and the bf16
# pages are faster); the tests run small batches, so they lower the gate: every complete page of every test is then read
# in its sealed form, and every parity test doubles as a test of that path.
os.environ.setdefault("MTTS_KV_PACK_MIN", "0")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN
