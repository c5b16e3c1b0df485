"""World-size-2 gloo test of the query-sharding path: shard ranges, CSR slicing and the single all-gather of the
top-k rows.  No GPU here, so each rank's shard is computed by the oracle (test infrastructure) -- the code under test
is doppel-speller_amd/distributed.py, which never computes anything itself."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_shard_ranges_cover_everything():
    from doppel_speller_amd.distributed import shard_range, shard_sizes
    for n in (0, 1, 7, 8, 100000, 100003):
        for world in (1, 2, 3, 8):
            ranges = [shard_range(n, r, world) for r in range(world)]
            assert ranges[0][0] == 0 and ranges[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(ranges, ranges[1:]))
            assert max(shard_sizes(n, world)) - min(shard_sizes(n, world)) <= 1


def test_slice_queries_round_trip():
    from doppel_speller_amd import synth
    from doppel_speller_amd.distributed import shard_range, slice_queries
    w = synth.make_workload(2000, 101, seed=5)
    pieces = [slice_queries(w.q_rowptr, w.q_cols, w.q_maxint, *shard_range(101, r, 3)) for r in range(3)]
    assert np.array_equal(np.concatenate([p[1] for p in pieces]), w.q_cols)
    assert np.array_equal(np.concatenate([p[2] for p in pieces]), w.q_maxint)
    assert all(p[0][0] == 0 and p[0][-1] == p[1].shape[0] for p in pieces)


def _worker(rank, world, port, n_queries, result_path):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    from doppel_speller_amd import synth
    from doppel_speller_amd.distributed import gather_rows, shard_range, slice_queries
    from oracle import oracle
    dist.init_process_group("gloo", rank=rank, world_size=world)
    w = synth.make_workload(3000, n_queries, seed=11)
    begin, end = shard_range(n_queries, rank, world)
    rowptr, cols, maxint = slice_queries(w.q_rowptr, w.q_cols, w.q_maxint, begin, end)
    local = oracle.jaccard_topk(w.rowptr, w.truth_idx, w.idf32, w.sums32, rowptr, cols, maxint, 10)
    gathered = gather_rows(torch.from_numpy(local), n_queries)
    if rank == 0:
        np.save(result_path, gathered.numpy())
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("n_queries", [64, 65])
def test_two_rank_gather_matches_single_process(tmp_path, n_queries):
    import torch.multiprocessing as mp
    from doppel_speller_amd import synth
    from oracle import oracle
    oracle.build()
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    result = str(tmp_path / "rows.npy")
    mp.spawn(_worker, args=(2, port, n_queries, result), nprocs=2, join=True)
    w = synth.make_workload(3000, n_queries, seed=11)
    expected = oracle.jaccard_topk(w.rowptr, w.truth_idx, w.idf32, w.sums32, w.q_rowptr, w.q_cols, w.q_maxint, 10)
    assert np.array_equal(np.load(result), expected)
