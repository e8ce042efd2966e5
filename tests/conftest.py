import os
import sys

os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")   # before anything initialises the HIP runtime (include/m17hip.h, m17hip_advice)

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
for p in (HERE, ROOT, os.path.join(ROOT, "m17-cxx-demod_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)


try:   # torch first: it carries its own HIP runtime, and a process that has initialised the GPU through the system one (libm17hip.so)
    import torch  # noqa: F401  before importing torch finds "no HIP GPUs" in torch afterwards (seen when a single GPU test file is run)
except Exception:   # pragma: no cover
    pass


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden():
    import json

    import numpy as np

    g = dict(np.load(os.path.join(HERE, "golden", "ref_vectors.npz")))
    g["kat"] = json.load(open(os.path.join(HERE, "golden", "kat_vectors.json")))
    return g
