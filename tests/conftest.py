import json
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


def load_golden(name):
    """Load tests/golden/<name>.npz -> (dict of arrays, meta dict)."""
    z = np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)
    d = {k: z[k] for k in z.files if k != "meta"}
    meta = json.loads(bytes(z["meta"]).decode())
    return d, meta


def golden_names(prefix=""):
    return sorted(f[:-4] for f in os.listdir(GOLDEN) if f.endswith(".npz") and f.startswith(prefix))


def rel_err(x, ref, atol=1e-6):
    """max |x-ref| / max(max |ref|, atol)  (tensor-scale relative error; atol guards the
    exactly-zero gradients of e.g. the N=1 masked case)."""
    x = np.asarray(x, dtype=np.float64)
    ref = np.asarray(ref, dtype=np.float64)
    return float(np.abs(x - ref).max() / max(np.abs(ref).max(), atol))


@pytest.fixture(scope="session")
def has_gpu():
    import torch
    return torch.cuda.is_available()
