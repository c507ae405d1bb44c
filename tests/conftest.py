import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def goldens():
    import json

    import numpy as np

    z = np.load(os.path.join(ROOT, "tests", "golden", "np_goldens.npz"))
    meta = json.loads(bytes(z["meta"]).decode())
    return z, meta


@pytest.fixture(scope="session")
def known_answers():
    import json

    with open(os.path.join(ROOT, "tests", "golden", "reference_known_answers.json")) as f:
        return json.load(f)


def rms(a):
    import numpy as np

    a = np.asarray(a, dtype=np.float64)
    return float(np.sqrt(np.mean(a * a))) if a.size else 0.0
