"""The three known-answer vectors the reference's OWN tests hold (doppelspeller/tests/test_common.py:16-28), kept as data in
tests/golden/reference_tests.json (captured by tests/golden/make_golden_transform.py from the reference's functions, asserted
there against the values its tests expect): transform_title, the words counter (a word repeated inside a title counts once --
the semantics behind get_truth_words_counts, row a7) and idf_word = ln(3 / 2) = 0.40547 (the idf_s formula of
feature_engineering.py:153)."""
import json
import math
import os

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


def _vectors():
    with open(os.path.join(HERE, "golden", "reference_tests.json"), encoding="utf-8") as handle:
        return json.load(handle)


def test_transform_title_known_answer():
    import doppel_speller_amd as ds
    from oracle import oracle
    v = _vectors()["transform_title"]
    assert v["transformed"] == "lkjblksd skjasl dfkjf 88 ggdjsdkj sdsd sdi d k bkjh77asda33"   # test_common.py:19
    assert oracle.transform_title(v["title"]) == v["transformed"]
    assert ds.transform_title(v["title"]) == v["transformed"]                  # the product's one-title form
    assert ds.transform_titles([v["title"], "x", v["title"]]) == [v["transformed"], ds.transform_title("x"), v["transformed"]]   # ds_transform_titles


def test_words_counter_known_answer():
    """get_words_counter (common.py:140-142) through ds_truth_word_counts: the count of a title's i-th word is the number of
    TITLES that hold the word."""
    from doppel_speller_amd.feature_engineering import get_truth_words_counts, truth_word_counts
    v = _vectors()["words_counter"]
    assert v["counts"] == {"first": 2, "second": 1, "third": 1, "fifth": 1}  # test_common.py:23
    titles = [" ".join(words) for words in v["titles_as_words"]]
    raw = np.frombuffer("".join(titles).encode("ascii"), dtype=np.uint8)
    offsets = np.concatenate(([0], np.cumsum([len(t) for t in titles]))).astype(np.int64)
    counts = truth_word_counts(raw, offsets, separators=[ord(" ")])
    for row, words in enumerate(v["titles_as_words"]):
        expected = [v["counts"][word] for word in words]
        assert counts[row, :len(words)].tolist() == expected
        assert not counts[row, len(words):].any()
        assert np.array_equal(counts[row], get_truth_words_counts(titles[row], v["counts"]))


def test_idf_word_known_answer_on_the_host():
    v = _vectors()["idf_word"]
    assert round(v["idf"], 5) == 0.40547                                        # test_common.py:28
    assert v["idf"] == math.log(v["number_of_titles"] / v["count"])            # common.py:158
    from oracle import oracle
    from doppel_speller_amd.feature_engineering import encode_titles
    enc, lengths = encode_titles(["first", "first"])
    counts = np.zeros((1, 15), dtype=np.uint32)
    counts[0, 0] = v["count"]
    features = oracle.construct_features(lengths[:1], lengths[1:], enc[:1], enc[1:], counts, 1, v["number_of_titles"])
    assert features[0, 36] == np.float32(v["idf"])                              # idf_s[0]: feature 6 + 15 + 15


@pytest.mark.gpu
def test_idf_word_known_answer_on_the_gpu():
    """idf_s of construct_features (feature_engineering.py:153) with number_of_truth_titles = 3 and a word count of 2:
    float32(ln(3 / 2)), through the 9-argument entry (the kernel's own log) and the indexed entry (the truth records)."""
    import doppel_speller_amd as ds
    from doppel_speller_amd.feature_engineering import TitleTable, construct_features_indexed, encode_titles
    v = _vectors()["idf_word"]
    enc, lengths = encode_titles(["first", "first second"])
    counts = np.zeros((1, 15), dtype=np.uint32)
    counts[0, :2] = (v["count"], 1)
    expected = np.float32(v["idf"])
    assert round(float(expected), 5) == 0.40547
    features = np.zeros((1, ds.FEATURES_COUNT), dtype=np.float32)
    ds.construct_features(lengths[:1], lengths[1:], enc[:1], enc[1:], counts, ds.SPACE_CODE, v["number_of_titles"],
                          np.zeros(ds.FEATURES_COUNT, dtype=np.uint8), features)
    assert features[0, 36] == expected and features[0, 37] == np.float32(math.log(3.0))
    queries, truth = TitleTable(enc[:1], lengths[:1]), TitleTable(enc[1:], lengths[1:], counts)
    indexed = construct_features_indexed(queries, truth, [0], [0], ds.SPACE_CODE, v["number_of_titles"])
    assert np.array_equal(indexed.view(np.uint32), features.view(np.uint32))
