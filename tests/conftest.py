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


def golden_frames(g):
    """(data, truth_data, vocabulary) for MatchMaker from a captured fixture: the n-gram column holds LISTS in the
    iteration order the reference's sets had when the fixture was captured (`truth_order_rank`), so the float32 sums
    of match_maker.py:174 are reproduced exactly whatever PYTHONHASHSEED the test process runs under."""
    import pandas as pd
    vocabulary = [str(v) for v in g["vocab"]]
    column = {n_gram: i for i, n_gram in enumerate(vocabulary)}
    n_grams = lambda title: {title[i:i + 3] for i in range(len(title) - 2)}
    ranks, at, ordered = g["truth_order_rank"], 0, []
    for title in g["truth_titles"]:
        ascending = sorted(n_grams(str(title)), key=column.get)
        ordered.append([ascending[r] for r in ranks[at:at + len(ascending)]])
        at += len(ascending)
    assert at == ranks.shape[0]
    truth = pd.DataFrame({"title_id": g["title_id"].astype(np.int64), "n_grams": ordered})
    data = pd.DataFrame({"n_grams": [sorted(n_grams(str(t))) for t in g["query_titles"]]})
    return data, truth, vocabulary


@pytest.fixture(scope="session")
def golden_match_maker_full():
    """The whole example truth set (30,000 rows = two score tiles) x 1,000 queries, captured from the reference."""
    g = dict(np.load(os.path.join(GOLDEN, "match_maker_30000x1000.npz"), allow_pickle=False))
    from doppel_speller_amd.match_maker import MatchMaker
    data, truth, vocabulary = golden_frames(g)
    g.update(MatchMaker.host_arrays(data, truth, vocabulary))
    return g


@pytest.fixture(scope="session")
def golden_features():
    return dict(np.load(os.path.join(GOLDEN, "construct_features_400.npz"), allow_pickle=False))


@pytest.fixture(scope="session")
def golden_features_edge():
    """construct_features of the reference on 126 pairs the example data does not contain (spaces at both ends, runs of
    spaces, > 15 words, one-character and 254-character titles ...), for three values of n_truth."""
    return dict(np.load(os.path.join(GOLDEN, "construct_features_edge.npz"), allow_pickle=False))


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
