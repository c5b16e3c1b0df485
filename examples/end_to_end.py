#!/usr/bin/env python3
"""The whole path of doppel-speller's `generate-predictions` on the GPU, from raw titles to match probabilities:

    transform_title -> MatchMaker (index build + Jaccard top-k) -> fuzzy close matches -> construct_features
                    -> tree ensemble

using only this package (reference: doppelspeller/predict.py:107-262).  Data and model are synthetic stand-ins: the
example data set and the pickled xgboost model of the reference are not part of this repository.

    python examples/end_to_end.py [n_truth] [n_queries] [top_n]
"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import doppel_speller_amd as ds  # noqa: E402
from doppel_speller_amd import synth  # noqa: E402


def main(n_truth=20000, n_queries=2000, top_n=10):
    workload = synth.make_workload(n_truth, n_queries, seed=11)
    raw_truth = [t.upper().replace(" ", "  ") + "." for t in synth._to_strings(workload.t_flat, workload.t_off)]
    raw_queries = [t.title() for t in synth._to_strings(workload.q_flat, workload.q_off)]

    t0 = time.perf_counter()
    truth = ds.transform_titles(raw_truth)                                   # common.py:20-47
    queries = ds.transform_titles(raw_queries)
    match_maker = ds.MatchMaker.from_titles(queries, truth, top_n)           # match_maker.py:84-109 (native build)
    rows = match_maker.get_closest_matches_batch()                           # match_maker.py:192-203, every query
    t1 = time.perf_counter()

    # encoded titles + truth word counts (feature_engineering.py:298-319), uploaded once
    words = {}
    for title in truth:
        for word in set(title.split()):
            words[word] = words.get(word, 0) + 1
    t_enc, t_len = ds.encode_titles(truth)
    q_enc, q_len = ds.encode_titles(queries)
    t_counts = np.stack([ds.get_truth_words_counts(title, words) for title in truth])
    truth_table = ds.TitleTable(t_enc, t_len, t_counts)
    query_table = ds.TitleTable(q_enc, q_len)

    ratios, best_row = ds.find_close_matches(query_table, truth_table, rows)  # predict.py:140-183
    pair_q = np.repeat(np.arange(len(queries), dtype=np.int32), top_n)
    features = ds.construct_features_indexed(query_table, truth_table, pair_q, rows.reshape(-1), ds.SPACE_CODE, len(truth))
    forest = synth.make_forest(n_trees=100)
    model = ds.ForestModel(forest["feature"], forest["threshold"], forest["yes"], forest["no"], forest["missing"],
                           forest["tree_offsets"], forest["n_features"], forest["base_margin"])
    probabilities = model.predict(features).reshape(len(queries), top_n)     # predict.py:229-234
    t2 = time.perf_counter()

    found = (best_row >= 0).sum()
    print(f"{len(truth)} truth titles, {len(queries)} queries, top-{top_n}")
    print(f"index build + top-k: {t1 - t0:.2f}s   close matches + features + model: {t2 - t1:.2f}s")
    print(f"fuzzy step alone decided {found} queries; feature matrix {features.shape}; "
          f"best model score per query: mean {probabilities.max(axis=1).mean():.3f}")
    return rows, best_row, features, probabilities


if __name__ == "__main__":
    main(*[int(a) for a in sys.argv[1:4]])
