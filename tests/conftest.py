import os
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

GOLDEN = os.path.join(REPO, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden():
    import numpy as np

    def load(name):
        return np.load(os.path.join(GOLDEN, name + ".npz"))
    return load


def rel_l2(a, b):
    import numpy as np
    a = np.asarray(a); b = np.asarray(b)
    n = np.linalg.norm(b.ravel())
    d = np.linalg.norm((a - b).ravel())
    return float(d / n) if n > 0 else float(d)


def ref_residual(a, b):
    """The reference's own snapshot metric sum((|F|-|D|)^2)/sum(|F|^2) (02_propagate.py:40-42)."""
    import numpy as np
    F, D = np.abs(np.asarray(a)), np.abs(np.asarray(b))
    return float(((F - D) ** 2).sum() / (F ** 2).sum())
