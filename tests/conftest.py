import glob
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN_DIR = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def golden_names():
    """GPTQ / AWQ / HQQ fixtures (tests/golden/make_goldens.py)."""
    return sorted(n for n in (os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLDEN_DIR, "*.npz")))
                  if not n.startswith("ort_"))


def ort_golden_names():
    """ORT / MatMulNBits blob-layout fixtures (tests/golden/make_goldens_ort.py)."""
    return sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLDEN_DIR, "ort_*.npz")))


def load_golden(name):
    d = dict(np.load(os.path.join(GOLDEN_DIR, name + ".npz"), allow_pickle=False))
    for k in ("layout",):
        d[k] = str(d[k])
    for k in ("bits", "groupsize", "K", "N", "compat"):
        d[k] = int(d[k])
    d["bias"] = d["bias"] if d["bias"].size else None
    return d


@pytest.fixture(params=golden_names())
def golden(request):
    return load_golden(request.param)
