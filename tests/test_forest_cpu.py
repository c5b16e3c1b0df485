"""Next row f-4 on the CPU: the xgboost JSON dump parser and the oracle's restatement of the prediction rule."""
import json
import math

import numpy as np


def _dump_tree(rng, n_features, depth, nodeid_counter=None):
    """A random tree in xgboost's `get_dump(dump_format='json')` layout (breadth-first node ids)."""
    nodes = [{"nodeid": 0, "depth": 0}]
    frontier = [nodes[0]]
    next_id = 1
    while frontier:
        node = frontier.pop(0)
        if node["depth"] >= depth or (node["depth"] > 0 and rng.rand() < 0.25):
            node["leaf"] = float(np.float32(rng.normal(0, 0.3)))
            continue
        node["split"] = f"f{rng.randint(n_features)}"
        node["split_condition"] = float(np.float32(rng.choice([rng.uniform(0, 100), rng.uniform(0, 12), 0.5, 1.5])))
        yes, no = next_id, next_id + 1
        next_id += 2
        node["yes"], node["no"] = yes, no
        node["missing"] = yes if rng.rand() < 0.5 else no
        children = [{"nodeid": yes, "depth": node["depth"] + 1}, {"nodeid": no, "depth": node["depth"] + 1}]
        node["children"] = children
        frontier.extend(children)
    return json.dumps(nodes[0])


def random_dump(seed, n_trees=40, n_features=66, depth=6):
    rng = np.random.RandomState(seed)
    return [_dump_tree(rng, n_features, depth) for _ in range(n_trees)]


def random_rows(seed, n, n_features=66):
    rng = np.random.RandomState(seed)
    rows = rng.uniform(0, 100, (n, n_features)).astype(np.float32)
    rows[rng.rand(n, n_features) < 0.3] = np.nan            # construct_features pads with NaN (:121-123)
    rows[:, :6] = rng.randint(0, 100, (n, 6))
    return rows


def _python_predict(forest, row):
    margin = np.float32(forest["base_margin"])
    for t in range(forest["tree_offsets"].shape[0] - 1):
        root = forest["tree_offsets"][t]
        node = root
        while forest["feature"][node] >= 0:
            value = row[forest["feature"][node]]
            if np.isnan(value):
                node = root + forest["missing"][node]
            elif value < forest["threshold"][node]:
                node = root + forest["yes"][node]
            else:
                node = root + forest["no"][node]
        margin = np.float32(margin + forest["threshold"][node])
    return margin


def test_dump_parser_and_oracle_agree_with_a_python_walk(oracle):
    from doppel_speller_amd.forest import ForestModel
    dump = random_dump(3, n_trees=12)
    forest = ForestModel.parse_xgboost_dump(dump, base_score=0.5)
    assert forest["tree_offsets"].shape[0] == 13 and forest["base_margin"] == 0.0
    limited = ForestModel.parse_xgboost_dump(dump, ntree_limit=5)
    assert limited["tree_offsets"].shape[0] == 6
    rows = random_rows(4, 200)
    margins, probabilities = oracle.forest_predict(forest, rows)
    expected = np.array([_python_predict(forest, row) for row in rows], dtype=np.float32)
    assert np.array_equal(margins.view(np.uint32), expected.view(np.uint32))
    reference = np.array([1.0 / (1.0 + math.exp(-float(m))) for m in margins])
    assert np.allclose(probabilities, reference, rtol=2e-7, atol=0)


def test_base_score_becomes_the_base_margin():
    from doppel_speller_amd.forest import ForestModel
    forest = ForestModel.parse_xgboost_dump(random_dump(1, n_trees=1), base_score=0.25)
    assert abs(forest["base_margin"] - math.log(0.25 / 0.75)) < 1e-12
