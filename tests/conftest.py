import glob
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "reference: needs /root/reference mounted (build container only)")


def golden(name):
    return dict(np.load(os.path.join(GOLDEN, name), allow_pickle=False))


def golden_files(pattern):
    return sorted(os.path.basename(p) for p in glob.glob(os.path.join(GOLDEN, pattern)))


def have_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    skip_ref = pytest.mark.skip(reason="/root/reference not mounted")
    have_ref = os.path.exists("/root/reference/PathNet_run.py")
    for item in items:
        if "reference" in item.keywords and not have_ref:
            item.add_marker(skip_ref)
