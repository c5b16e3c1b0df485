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


@pytest.fixture(scope="session")
def golden_match_maker():
    return dict(np.load(os.path.join(GOLDEN, "match_maker_5000x200.npz"), allow_pickle=False))


@pytest.fixture(scope="session")
def golden_features():
    return dict(np.load(os.path.join(GOLDEN, "construct_features_400.npz"), allow_pickle=False))


@pytest.fixture(scope="session")
def golden_kat():
    import json
    with open(os.path.join(GOLDEN, "kat.json")) as handle:
        return json.load(handle)


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle as module
    module.build()
    return module
