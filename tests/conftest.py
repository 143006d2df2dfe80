import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "gpu_slow: needs a real MI355X AND minutes of reference CPU time; run on demand (-m gpu_slow), log kept under profiles/")


def pytest_collection_modifyitems(config, items):
    """gpu_slow cases run only when the -m expression names them: the regular -m gpu run covers the same ground through the seeded
    fixtures (tests/test_gpu_gen_parity.py) without a minute and a half of reference CPU time on the GPU box."""
    if "gpu_slow" in (config.getoption("-m") or ""):
        return
    skip = pytest.mark.skip(reason="on demand: -m gpu_slow")
    for item in items:
        if item.get_closest_marker("gpu_slow"):
            item.add_marker(skip)


GOLDEN = os.path.join(ROOT, "tests", "golden")
