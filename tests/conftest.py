import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "gpu_slow: needs a real MI355X AND minutes of reference CPU time; run on demand (-m gpu_slow), log kept under profiles/")


GOLDEN = os.path.join(ROOT, "tests", "golden")
