"""the ctypes stub a maintainer of the reference would add (examples/reference_binding.py),
executed against the golden fixture -- rows from the stub == rows the reference produced."""
import importlib.util
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_integration_document_shows_the_tested_stub():
    with open(os.path.join(ROOT, "INTEGRATION.md")) as handle:
        document = handle.read()
    with open(os.path.join(ROOT, "examples", "reference_binding.py")) as handle:
        stub = handle.read()
    shown = re.search(r"```python\n(# doppelspeller/_amd.py.*?)```", document, re.S).group(1)
    assert shown == stub


class _ReferenceMatchMaker:
    """The four attributes of the reference's MatchMaker the stub reads (match_maker.py:99-107,180-181)."""

    def __init__(self, g):
        self.matrix_truth_non_zero_columns_and_values = [
            (g["truth_idx"][g["rowptr"][c]:g["rowptr"][c + 1]], None) for c in range(g["rowptr"].shape[0] - 1)]
        self.matrix_non_zero_columns = [g["q_cols"][g["q_rowptr"][q]:g["q_rowptr"][q + 1]]
                                        for q in range(g["q_rowptr"].shape[0] - 1)]
        self.sums_matrix_truth = g["sums32"]
        self._idf = g["idf32"]

    def _get_idf_given_index(self, index):
        return float(self._idf[index])


@pytest.mark.gpu
def test_stub_reproduces_the_reference_rows_and_features(golden_match_maker, golden_features):
    from doppel_speller_amd import _lib
    os.environ["DOPPEL_AMD_LIBRARY"] = _lib.library_path()
    spec = importlib.util.spec_from_file_location("reference_binding", os.path.join(ROOT, "examples", "reference_binding.py"))
    stub = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(stub)
    g = golden_match_maker
    index = stub.AmdIndex(_ReferenceMatchMaker(g))
    rows = index.top_rows(_ReferenceMatchMaker(g), list(range(200)), 10)
    assert np.array_equal(rows, g["rows_k10"])               # margin_ok is all true at k = 10
    with pytest.raises(Exception, match="top_matches.shape"):
        index.top_rows(_ReferenceMatchMaker(g), [0], 5001)
    f = golden_features
    response = np.zeros((f["title_len"].shape[0], 66), dtype=np.float32)
    stub.construct_features(f["title_len"], f["truth_len"], f["title_enc"], f["truth_enc"], f["counts"], f["space_code"],
                            f["n_truth"], np.zeros(66, np.uint8), response)
    assert np.array_equal(response[:, :51].view(np.uint32), f["features"][:, :51].view(np.uint32))
