"""Offline model (numpy, CPU): how many rows pass the collect sweep's register-level test -- i.e. become RAW ENTRIES that cost a
row-record gather each -- as a function of the information a posting carries, for the selection-wide skip set and for MaxScore per
tile (profiles/r04_tuning.txt, batches 8-9: the per-tile variant lost to a 2.4x flood of raw entries).

Per query (final thresholds, as in scripts/sim_tile_maxscore.py), for every row of the band's tiles that holds at least one ESSENTIAL
column:   pass  <=>  essential score + upper bound of the skipped columns' mass for this row  >=  coef * (lower bound of sums32 + maxint)
with      upper bound = sum over the skipped columns the row is KNOWN to hold (its membership bit travels with the posting: the B
                        densest columns of the index) + all skipped columns whose membership is not known (assumed present)
          lower bound of sums32 = the 8-bit float code of the index (4-bit exponent, 4-bit mantissa, truncated) or the exact value.
Fixed-point slack is ignored (it adds a few per cent to every variant alike).

usage: sim_posting_info.py <truth titles> <k> [queries]      (~2 min at 500k x 100 queries)"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
from doppel_speller_amd import synth  # noqa: E402

N, K = int(sys.argv[1]), int(sys.argv[2])
Q = int(sys.argv[3]) if len(sys.argv) > 3 else 100
TILE = 12288 if N <= 10_000_000 else 28672
w = synth.make_workload(N, 2000, seed=20260101)
rowptr, tidx, idf32, sums32 = (np.asarray(x) for x in (w.rowptr, w.truth_idx, w.idf32, w.sums32))
order = np.argsort(sums32, kind="stable")
pos = np.empty(N, np.int64)
pos[order] = np.arange(N)
sums_sorted = sums32[order]
ntiles = (N + TILE - 1) // TILE
tile_min = sums_sorted[np.arange(ntiles) * TILE]
tile_max = sums_sorted[np.minimum(N, (np.arange(ntiles) + 1) * TILE) - 1]
tile_of_row = pos // TILE
tile_min_of_row = tile_min[tile_of_row].astype(np.float64)   # no per-row sums information at all: the tile's smallest
df = np.diff(rowptr)
dense_rank = np.empty(len(df), np.int64)
dense_rank[np.argsort(-df, kind="stable")] = np.arange(len(df))          # 0 = the densest column of the index
has_sig = (dense_rank < 128) & (df * 256 >= N)


def code8_lower_bound(x):
    """decode(encode_sums8(x)) of csrc/ds_common.h: 4-bit exponent (2^-3 .. 2^12), 4-bit mantissa, truncated."""
    bits = x.astype(np.float32).view(np.uint32)
    exponent = (bits >> 23).astype(np.int64) - 124
    code = np.where(x < 0.125, 0, np.minimum(0xfe, (np.clip(exponent, 0, 15) << 4) | ((bits >> 19) & 0xf)))
    code = np.where(exponent > 15, 0xfe, code)
    out = ((((code >> 4) + 124).astype(np.uint32) << 23) | ((code & 0xf).astype(np.uint32) << 19)).view(np.float32)
    return np.where(code == 0, 0.0, out).astype(np.float64)


sums_lb8 = code8_lower_bound(sums32)
variants = [(b, s, t) for t in ("selection", "tile") for s in ("code8", "exact", "tilemin") for b in (8, 16, 32, 128)]
raw = {v: 0 for v in variants}
survivors = 0
essential_rows = {"selection": 0, "tile": 0}
rng = np.random.RandomState(1)
done = 0
for q in rng.choice(2000, Q, replace=False):
    cols = np.asarray(w.q_cols[w.q_rowptr[q]:w.q_rowptr[q + 1]])
    n = len(cols)
    if n == 0:
        continue
    maxint = float(w.q_maxint[q])
    o = np.argsort(idf32[cols], kind="stable")
    cs = cols[o]
    idf = idf32[cs].astype(np.float64)
    mass = np.cumsum(idf)
    member = np.zeros((n, N), dtype=bool)                       # member[i, row]: row holds the query's i-th column (ascending idf)
    for i, c in enumerate(cs):
        member[i, tidx[rowptr[c]:rowptr[c + 1]]] = True
    total = (member * idf[:, None]).sum(axis=0)
    jac = total / (sums32.astype(np.float64) + (maxint - total))
    kth = np.partition(jac, N - K)[N - K]
    if kth <= 0:
        continue
    done += 1
    cut = kth - 1e-5
    coef = cut / (1 + cut)
    band_tile = (tile_min * cut <= maxint) & (tile_max >= cut * maxint)
    in_band = band_tile[tile_of_row]
    sig_ok = np.cumprod(has_sig[cs]).astype(bool)

    def skip_for(pre):
        ok = (mass < pre) & sig_ok
        return int(np.argmin(ok)) if not ok.all() else n
    survivors += int(((total >= coef * (sums32 + maxint)) & in_band).sum())
    for scheme in ("selection", "tile"):
        if scheme == "selection":
            skip_row = np.full(N, skip_for(coef * (float(sums_sorted[0]) + maxint)))
        else:
            skip_tile = np.array([skip_for(coef * (float(tile_min[t]) + maxint)) for t in range(ntiles)])
            skip_row = skip_tile[tile_of_row]
        ranks = np.arange(n)[:, None]
        skipped = ranks < skip_row[None, :]                       # [column rank, row]: is the column skipped in the row's tile
        essential_score = (member & ~skipped) * idf[:, None]
        s_ess = essential_score.sum(axis=0)
        touched = (s_ess > 0) & in_band
        essential_rows[scheme] += int(touched.sum())
        for bits in (8, 16, 32, 128):
            known = dense_rank[cs] < bits                            # membership of these columns travels with the posting
            bound = ((skipped & (member | ~known[:, None])) * idf[:, None]).sum(axis=0)
            for sums_kind in ("code8", "exact", "tilemin"):
                lower = sums_lb8 if sums_kind == "code8" else (sums32.astype(np.float64) if sums_kind == "exact" else tile_min_of_row)
                raw[(bits, sums_kind, scheme)] += int((touched & (s_ess + bound >= coef * (lower + maxint))).sum())
    del member
print(f"N={N} k={K} queries={done}: rows holding an essential column in the band per query: selection-wide skip "
      f"{essential_rows['selection'] / done:.0f}, skip per tile {essential_rows['tile'] / done:.0f}; rows that truly pass the loose test "
      f"(survivors of the refinement) {survivors / done:.1f}")
print(f"{'raw entries per query':34s}" + "".join(f"{'B=' + str(b):>10s}" for b in (8, 16, 32, 128)))
for scheme in ("selection", "tile"):
    for sums_kind in ("code8", "exact", "tilemin"):
        print(f"skip per {scheme:9s} sums {sums_kind:7s}  " + "".join(f"{raw[(b, sums_kind, scheme)] / done:10.1f}" for b in (8, 16, 32, 128)))
