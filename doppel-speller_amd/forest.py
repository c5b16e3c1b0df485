"""Next row f-4: the tree ensemble of predict.py:229-234 applied to the feature matrix on the GPU.

The reference pickles an `xgboost.Booster` (train.py:135, predict.py:80-82) and calls
`model.predict(xgb.DMatrix(features), ntree_limit=model.best_ntree_limit)`.  xgboost is not part of the reference tree;
a maintainer exports the booster once with `model.get_dump(dump_format='json')` (a list with one JSON string per tree)
and loads it here with `ForestModel.from_xgboost_dump(...)`.  Prediction follows xgboost's published rule for
`binary:logistic` (parity unpinned against the library itself; the GPU tests compare margins bit-for-bit with a CPU restatement of the same rule).
"""
import ctypes
import json
import math

import numpy as np

from . import _lib


def _ptr(array):
    return array.ctypes.data_as(ctypes.c_void_p)


class ForestModel:
    def __init__(self, feature, threshold, yes, no, missing, tree_offsets, n_features, base_margin=0.0, device=0):
        self.arrays = dict(feature=np.ascontiguousarray(feature, dtype=np.int32),
                           threshold=np.ascontiguousarray(threshold, dtype=np.float32),
                           yes=np.ascontiguousarray(yes, dtype=np.int32), no=np.ascontiguousarray(no, dtype=np.int32),
                           missing=np.ascontiguousarray(missing, dtype=np.int32),
                           tree_offsets=np.ascontiguousarray(tree_offsets, dtype=np.int64),
                           base_margin=float(base_margin))
        self.n_features = int(n_features)
        self.n_trees = self.arrays["tree_offsets"].shape[0] - 1
        self.device = device
        self.handle = ctypes.c_void_p()
        a = self.arrays
        _lib.check(_lib.lib().ds_forest_create(_ptr(a["feature"]), _ptr(a["threshold"]), _ptr(a["yes"]), _ptr(a["no"]),
                                               _ptr(a["missing"]), _ptr(a["tree_offsets"]), self.n_trees,
                                               self.n_features, ctypes.c_float(a["base_margin"]), device,
                                               ctypes.byref(self.handle)), "ds_forest_create")

    @staticmethod
    def parse_xgboost_dump(trees, ntree_limit=None, base_score=0.5):
        """`Booster.get_dump(dump_format='json')` -> the flat arrays (feature names 'f<index>')."""
        if ntree_limit:
            trees = trees[:ntree_limit]                       # predict.py:232 ntree_limit=best_ntree_limit
        feature, threshold, yes, no, missing, offsets = [], [], [], [], [], [0]
        for text in trees:
            root = json.loads(text) if isinstance(text, str) else text
            nodes = {}
            stack = [root]
            while stack:
                node = stack.pop()
                nodes[int(node["nodeid"])] = node
                stack.extend(node.get("children", ()))
            size = max(nodes) + 1
            f = np.full(size, -1, np.int32)
            t = np.zeros(size, np.float32)
            y = np.zeros(size, np.int32)
            n = np.zeros(size, np.int32)
            m = np.zeros(size, np.int32)
            for nodeid, node in nodes.items():
                if "leaf" in node:
                    t[nodeid] = np.float32(node["leaf"])
                else:
                    split = node["split"]
                    f[nodeid] = int(split[1:]) if isinstance(split, str) else int(split)
                    t[nodeid] = np.float32(node["split_condition"])
                    y[nodeid], n[nodeid], m[nodeid] = int(node["yes"]), int(node["no"]), int(node["missing"])
            feature.append(f); threshold.append(t); yes.append(y); no.append(n); missing.append(m)
            offsets.append(offsets[-1] + size)
        cat = lambda parts, dtype: np.concatenate(parts).astype(dtype) if parts else np.zeros(0, dtype)
        base_margin = math.log(base_score / (1.0 - base_score))
        return dict(feature=cat(feature, np.int32), threshold=cat(threshold, np.float32), yes=cat(yes, np.int32),
                    no=cat(no, np.int32), missing=cat(missing, np.int32),
                    tree_offsets=np.array(offsets, dtype=np.int64), base_margin=base_margin)

    @staticmethod
    def parse_xgboost_model_json(model, ntree_limit=None):
        """`Booster.save_model('model.json')` (xgboost >= 1.0) -> the flat arrays.  Unlike the text dump, this format
        carries every split condition and leaf value as the exact float32 the booster holds."""
        if isinstance(model, (str, bytes)):
            model = json.loads(model)
        learner = model["learner"]
        trees = learner["gradient_booster"]["model"]["trees"]
        if ntree_limit:
            trees = trees[:ntree_limit]
        feature, threshold, yes, no, missing, offsets = [], [], [], [], [], [0]
        for tree in trees:
            left = np.asarray(tree["left_children"], dtype=np.int32)
            right = np.asarray(tree["right_children"], dtype=np.int32)
            leaf = left < 0
            f = np.where(leaf, -1, np.asarray(tree["split_indices"], dtype=np.int32)).astype(np.int32)
            t = np.asarray(tree["split_conditions"], dtype=np.float32)      # leaves: the leaf value
            default_left = np.asarray(tree["default_left"], dtype=bool)
            feature.append(f); threshold.append(t)
            yes.append(np.where(leaf, 0, left)); no.append(np.where(leaf, 0, right))
            missing.append(np.where(leaf, 0, np.where(default_left, left, right)))
            offsets.append(offsets[-1] + left.shape[0])
        cat = lambda parts, dtype: np.concatenate(parts).astype(dtype) if parts else np.zeros(0, dtype)
        base_score = float(learner["learner_model_param"]["base_score"])
        return dict(feature=cat(feature, np.int32), threshold=cat(threshold, np.float32), yes=cat(yes, np.int32),
                    no=cat(no, np.int32), missing=cat(missing, np.int32),
                    tree_offsets=np.array(offsets, dtype=np.int64),
                    base_margin=math.log(base_score / (1.0 - base_score)))

    @classmethod
    def from_xgboost_model_json(cls, model, n_features, ntree_limit=None, device=0):
        a = cls.parse_xgboost_model_json(model, ntree_limit)
        return cls(a["feature"], a["threshold"], a["yes"], a["no"], a["missing"], a["tree_offsets"], n_features,
                   a["base_margin"], device)

    @classmethod
    def from_xgboost_dump(cls, trees, n_features, ntree_limit=None, base_score=0.5, device=0):
        a = cls.parse_xgboost_dump(trees, ntree_limit, base_score)
        return cls(a["feature"], a["threshold"], a["yes"], a["no"], a["missing"], a["tree_offsets"], n_features,
                   a["base_margin"], device)

    def predict(self, rows, output_margin=False):
        """model.predict(xgb.DMatrix(rows)) for a host float32[n, n_features] matrix."""
        rows = np.ascontiguousarray(rows, dtype=np.float32)
        assert rows.ndim == 2 and rows.shape[1] == self.n_features
        out = np.empty(rows.shape[0], dtype=np.float32)
        null = ctypes.c_void_p(0)
        _lib.check(_lib.lib().ds_forest_predict(self.handle, _ptr(rows), rows.shape[0],
                                                _ptr(out) if output_margin else null,
                                                null if output_margin else _ptr(out)), "ds_forest_predict")
        return out

    def predict_device(self, d_rows, n, d_margins=None, d_probabilities=None, stream=None):
        as_ptr = lambda x: ctypes.c_void_p(0) if x is None else (x if isinstance(x, ctypes.c_void_p) else ctypes.c_void_p(int(x)))
        _lib.check(_lib.lib().ds_forest_predict_device(self.handle, as_ptr(d_rows), n, as_ptr(d_margins),
                                                       as_ptr(d_probabilities), ctypes.c_void_p(stream or 0)),
                   "ds_forest_predict_device")

    def close(self):
        if self.handle:
            _lib.lib().ds_forest_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
