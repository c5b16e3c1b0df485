import sys, numpy as np
sys.path.insert(0, '.')
from oracle import oracle
import doppel_speller_amd as ds
rng = np.random.RandomState(int(sys.argv[1]) if len(sys.argv) > 1 else 1)
LENS = [0, 1, 2, 3, 5, 17, 40, 63, 64, 65, 100, 127, 128, 129, 200, 254, 255]
def make(style):
    n = LENS[rng.randint(len(LENS))] if rng.rand() < 0.6 else rng.randint(0, 256)
    if style == 0:   # words separated by single spaces, small alphabet
        s = rng.randint(2, 8, n)
        s[rng.rand(n) < 0.2] = 1
    elif style == 1:  # many spaces, runs of spaces, leading / trailing
        s = rng.randint(1, 4, n)
    elif style == 2:  # wide alphabet incl. codes >= 64
        s = rng.randint(0, 256, n)
    elif style == 3:  # all spaces / all same
        s = np.full(n, rng.choice([1, 7, 200]))
    else:            # codes < 64, long words
        s = rng.randint(2, 64, n); s[rng.rand(n) < 0.05] = 1
    return s.astype(np.uint8)
total_bad = 0
for it in range(6):
    n = 4000
    q_enc = np.zeros((n, 255), np.uint8); t_enc = np.zeros((n, 255), np.uint8)
    q_len = np.zeros(n, np.uint8); t_len = np.zeros(n, np.uint8)
    for i in range(n):
        style = rng.randint(5)
        t = make(style)
        if rng.rand() < 0.5 and len(t) > 0:
            q = t.copy()
            for _ in range(rng.randint(0, 5)):
                at = rng.randint(len(q)) if len(q) else 0
                if rng.rand() < 0.5 and len(q) > 1: q = np.delete(q, at)
                elif len(q) < 255: q = np.insert(q, at, rng.randint(1, 70))
        else:
            q = make(rng.randint(5))
        q_enc[i, :len(q)] = q; q_len[i] = len(q); t_enc[i, :len(t)] = t; t_len[i] = len(t)
    counts = rng.randint(0, 5, (n, 15)).astype(np.uint32) * rng.randint(1, 20000, (n, 15)).astype(np.uint32)
    n_truth = [1, 7, 30000, 50000000][it % 4]
    feats = np.zeros((n, 66), np.float32)
    with np.errstate(all="ignore"):
        ds.construct_features(q_len, t_len, q_enc, t_enc, counts, np.uint8(1), np.uint32(n_truth), None, feats)
        exp = oracle.construct_features(q_len, t_len, q_enc, t_enc, counts, 1, n_truth)
    bad = np.nonzero((feats.view(np.uint32) != exp.view(np.uint32)).any(axis=1))[0]
    total_bad += bad.shape[0]
    print("iteration", it, "n_truth", n_truth, "bad", bad.shape[0], flush=True)
    for b in bad[:3]:
        cols = np.nonzero(feats[b].view(np.uint32) != exp[b].view(np.uint32))[0]
        print("  pair", b, "lens", q_len[b], t_len[b], "cols", cols[:8], feats[b][cols[:4]], exp[b][cols[:4]], "q", q_enc[b, :min(12, q_len[b])], "t", t_enc[b, :min(12, t_len[b])])
print("TOTAL BAD", total_bad)
