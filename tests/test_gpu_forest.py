"""Next row f-4 on the GPU: the tree-ensemble kernel against the oracle restatement (xgboost itself is outside the
reference tree: parity unpinned against the library, bit-exact margins against oracle/doppel_oracle.c)."""
import numpy as np
import pytest

from test_forest_cpu import random_dump, random_rows

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("n_trees,depth,n", [(1, 1, 5), (60, 6, 10000), (300, 4, 3000)])
def test_forest_kernel_against_oracle(oracle, n_trees, depth, n):
    import doppel_speller_amd as ds
    dump = random_dump(n_trees + depth, n_trees=n_trees, depth=depth)
    model = ds.ForestModel.from_xgboost_dump(dump, 66)
    rows = random_rows(n, n)
    margins, probabilities = oracle.forest_predict(model.arrays, rows)
    assert np.array_equal(model.predict(rows, output_margin=True).view(np.uint32), margins.view(np.uint32))
    got = model.predict(rows)
    assert np.allclose(got, probabilities, rtol=3e-7, atol=0)       # float32 exp: device vs libm, <= 2 ulp
    assert ((got > 0) & (got < 1)).all()


def test_ntree_limit_and_base_score(oracle):
    import doppel_speller_amd as ds
    dump = random_dump(11, n_trees=30)
    rows = random_rows(12, 500)
    limited = ds.ForestModel.from_xgboost_dump(dump, 66, ntree_limit=7, base_score=0.2)
    margins, _ = oracle.forest_predict(limited.arrays, rows)
    assert limited.n_trees == 7
    assert np.array_equal(limited.predict(rows, output_margin=True).view(np.uint32), margins.view(np.uint32))


def test_bad_forests_are_rejected():
    import doppel_speller_amd as ds
    offsets = np.array([0, 3], np.int64)
    feature = np.array([0, -1, -1], np.int32)
    threshold = np.array([0.5, 1.0, 2.0], np.float32)
    ok = dict(yes=np.array([1, 0, 0], np.int32), no=np.array([2, 0, 0], np.int32), missing=np.array([1, 0, 0], np.int32))
    ds.ForestModel(feature, threshold, ok["yes"], ok["no"], ok["missing"], offsets, 66)
    with pytest.raises(ds.DoppelError):                                   # child before its parent: a cycle
        ds.ForestModel(feature, threshold, np.array([0, 0, 0], np.int32), ok["no"], ok["missing"], offsets, 66)
    with pytest.raises(ds.DoppelError):                                   # feature index out of range
        ds.ForestModel(np.array([70, -1, -1], np.int32), threshold, ok["yes"], ok["no"], ok["missing"], offsets, 66)


def test_predictions_inside_the_device_pipeline(oracle):
    import doppel_speller_amd as ds
    from doppel_speller_amd import synth
    w = synth.make_workload(20000, 500, seed=41)
    pipeline = ds.CandidatePipeline(w, 10)
    model = ds.ForestModel.from_xgboost_dump(random_dump(5, n_trees=80), ds.FEATURES_COUNT)
    pipeline.step()
    pipeline.enqueue_predict(model)
    assert pipeline.sync()["error_queries"] == 0
    features = pipeline.features()
    _, probabilities = oracle.forest_predict(model.arrays, features)
    assert np.allclose(pipeline.predictions().reshape(-1), probabilities, rtol=3e-7, atol=0)
