import sys, numpy as np
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import test_gpu_jaccard as T
from oracle import oracle
import doppel_speller_amd as ds
rng = np.random.RandomState(5)
p = T._random_problem(rng, 60000, 700, 200, heavy=12)
# duplicate some columns inside every query (dense ones included)
q_rowptr, q_cols, q_maxint = [0], [], []
for q in range(200):
    c = p["q_cols"][p["q_rowptr"][q]:p["q_rowptr"][q + 1]]
    extra = np.concatenate((c, c[rng.rand(c.shape[0]) < 0.4], np.arange(0, 12)[rng.rand(12) < 0.5], np.arange(0, 12)[rng.rand(12) < 0.5]))
    rng.shuffle(extra)
    q_cols.append(extra.astype(np.int32)); q_rowptr.append(q_rowptr[-1] + extra.shape[0])
    q_maxint.append(float(sum(float(p["idf32"][g]) for g in extra)))
p["q_rowptr"] = np.array(q_rowptr, np.int64); p["q_cols"] = np.concatenate(q_cols); p["q_maxint"] = np.array(q_maxint)
index = ds.TruthIndex(p["rowptr"], p["truth_idx"], p["idf32"], p["sums32"])
for k in (1, 10, 50):
    got = index.top_k(p["q_rowptr"], p["q_cols"], p["q_maxint"], k)
    exp = oracle.jaccard_topk(p["rowptr"], p["truth_idx"], p["idf32"], p["sums32"], p["q_rowptr"], p["q_cols"], p["q_maxint"], k)
    bad = np.nonzero((got != exp).any(axis=1))[0]
    print("k", k, "bad", bad.shape[0], index.sync()["dense_queries"], index.status(200)[bad[:10]] if bad.size else "")
