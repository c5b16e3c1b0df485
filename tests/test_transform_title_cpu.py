"""transform_title (common.py:20-47): the Python restatement and the native batch form against vectors captured from
the reference's own function (tests/golden/make_golden_transform.py) -- parity pinned."""
import json
import os

HERE = os.path.dirname(os.path.abspath(__file__))


def _vectors():
    with open(os.path.join(HERE, "golden", "transform_title.json"), encoding="utf-8") as handle:
        return json.load(handle)


def test_python_restatement_matches_the_reference():
    from oracle import oracle
    import doppel_speller_amd as ds
    vectors = _vectors()
    assert len(vectors) >= 300
    for vector in vectors:
        assert oracle.transform_title(vector["title"]) == vector["transformed"], vector["title"]
    for vector in vectors[:40]:   # the product's one-title form goes through the native batch path
        assert ds.transform_title(vector["title"]) == vector["transformed"], vector["title"]


def test_native_batch_matches_the_reference():
    import doppel_speller_amd as ds
    vectors = _vectors()
    got = ds.transform_titles([v["title"] for v in vectors])
    assert got == [v["transformed"] for v in vectors]
    assert ds.transform_titles([]) == []


def test_titles_feed_the_native_index_build():
    """raw titles -> transform_titles -> NativeProblem: the whole host chain without per-title Python objects."""
    import doppel_speller_amd as ds
    raw = [v["title"] for v in _vectors()[40:240]]
    titles = ds.transform_titles(raw)
    problem = ds.NativeProblem(titles, titles[:50])
    assert problem.n_truth == 200 and problem.n_queries == 50 and problem.n_columns > 100
