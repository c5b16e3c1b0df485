"""CPU tests of the query-sharding path: shard ranges, CSR slicing, the TCP rendezvous and the single gather of the
top-k rows with the injectable host communicator (world sizes 2 and 3, fresh child processes).  No GPU here, so each
rank's shard is computed by the oracle (test infrastructure) -- the code under test is
doppel-speller_amd/distributed.py, which never computes anything itself.  The RCCL communicator has the same
`all_gather` interface and is exercised on the GPU box (tests/test_gpu_distributed.py)."""
import multiprocessing
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


def test_product_package_does_not_import_torch():
    """north_star: host code is Python over a thin C ABI, no PyTorch -- not even for the multi-GPU plumbing."""
    import subprocess
    script = ("import sys; sys.path.insert(0, %r); import doppel_speller_amd, doppel_speller_amd.distributed; "
              "assert 'torch' not in sys.modules, 'torch was imported'; print('clean')" % ROOT)
    result = subprocess.run([sys.executable, "-c", script], capture_output=True, text=True, timeout=120)
    assert result.returncode == 0 and "clean" in result.stdout, result.stderr[-1000:]
    for name in os.listdir(os.path.join(ROOT, "doppel-speller_amd")):
        if name.endswith(".py"):
            with open(os.path.join(ROOT, "doppel-speller_amd", name)) as handle:
                assert "import torch" not in handle.read(), name
    with open(os.path.join(ROOT, "bench.py")) as handle:
        assert "import torch" not in handle.read()


def _worker(rank, world, port, n_queries, result_path):
    sys.path.insert(0, ROOT)
    from doppel_speller_amd import synth
    from doppel_speller_amd.distributed import HostCommunicator, Rendezvous, RowGather, shard_range, slice_queries
    from oracle import oracle
    rendezvous = Rendezvous(rank, world, "127.0.0.1", port, timeout=120.0)
    assert rendezvous.broadcast_bytes(b"unique-id" if rank == 0 else None) == b"unique-id"
    assert rendezvous.max(float(rank)) == float(world - 1)
    w = synth.make_workload(3000, n_queries, seed=11)
    begin, end = shard_range(n_queries, rank, world)
    rowptr, cols, maxint = slice_queries(w.q_rowptr, w.q_cols, w.q_maxint, begin, end)
    local = oracle.jaccard_topk(w.rowptr, w.truth_idx, w.idf32, w.sums32, rowptr, cols, maxint, 10)
    gather = RowGather(HostCommunicator(rendezvous), n_queries, 10)
    for _ in range(2):  # buffers are reused from call to call
        gathered = gather.gather(local)
    rendezvous.barrier()
    np.save(f"{result_path}.{rank}.npy", gathered)
    rendezvous.close()


@pytest.mark.parametrize("world,n_queries", [(2, 64), (2, 65), (3, 100), (8, 1001)])   # 8 = the node the driver runs C4 / C5 on
def test_gather_matches_single_process(tmp_path, world, n_queries):
    from doppel_speller_amd import synth
    from oracle import oracle
    oracle.build()
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    result = str(tmp_path / "rows")
    context = multiprocessing.get_context("spawn")
    ranks = [context.Process(target=_worker, args=(r, world, port, n_queries, result)) for r in range(world)]
    for process in ranks:
        process.start()
    for process in ranks:
        process.join(300)
        assert process.exitcode == 0
    w = synth.make_workload(3000, n_queries, seed=11)
    expected = oracle.jaccard_topk(w.rowptr, w.truth_idx, w.idf32, w.sums32, w.q_rowptr, w.q_cols, w.q_maxint, 10)
    for rank in range(world):   # an all-gather: every rank holds all rows, in query order
        assert np.array_equal(np.load(f"{result}.{rank}.npy"), expected)


def _env_worker(rank, world, master_port, out_path):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(master_port))
    from doppel_speller_amd.distributed import Rendezvous
    rendezvous = Rendezvous.from_environment()
    parts = rendezvous.all_gather_bytes(bytes([rank]) * 3)
    rendezvous.barrier()
    with open(f"{out_path}.{rank}", "wb") as handle:
        handle.write(b"".join(parts))
    rendezvous.close()


def test_rendezvous_from_environment_when_the_preferred_port_is_taken(tmp_path):
    """MASTER_PORT + 1 is occupied by somebody else: rank 0 listens elsewhere and publishes the port in a file."""
    with socket.socket() as blocker:
        blocker.bind(("127.0.0.1", 0))
        blocker.listen(1)
        master_port = blocker.getsockname()[1] - 1
        context = multiprocessing.get_context("spawn")
        out = str(tmp_path / "parts")
        ranks = [context.Process(target=_env_worker, args=(r, 3, master_port, out)) for r in range(3)]
        for process in ranks:
            process.start()
        for process in ranks:
            process.join(120)
            assert process.exitcode == 0
    for rank in range(3):
        with open(f"{out}.{rank}", "rb") as handle:
            assert handle.read() == b"\x00\x00\x00\x01\x01\x01\x02\x02\x02"


def _free_port():
    with socket.socket() as probe:
        probe.bind(("127.0.0.1", 0))
        return probe.getsockname()[1]


def _rccl_failure_worker(rank, world, port, out_path):
    sys.path.insert(0, ROOT)
    from doppel_speller_amd.distributed import RcclCommunicator, Rendezvous
    rendezvous = Rendezvous(rank, world, port=port)
    try:
        RcclCommunicator(rendezvous, rank)
        outcome = "initialised"
    except Exception as error:  # noqa: BLE001 - the message is the result
        outcome = f"{type(error).__name__}: {error}"
    flags = rendezvous.all_gather_bytes(outcome.encode())   # the rendezvous is still in step on every rank
    with open(f"{out_path}.{rank}", "w") as handle:
        handle.write("\n".join(flag.decode() for flag in flags))
    rendezvous.close()


def test_rccl_initialisation_failure_raises_on_every_rank_without_hanging(tmp_path):
    """No GPU here: rank 0 cannot select a device, so it never creates a unique id.  It still takes part in the id
    broadcast (empty id): every rank raises and the rendezvous stays usable -- what bench.py's agreed fallback to the
    host gather relies on."""
    context = multiprocessing.get_context("spawn")
    port = _free_port()
    out = str(tmp_path / "outcome")
    ranks = [context.Process(target=_rccl_failure_worker, args=(r, 2, port, out)) for r in range(2)]
    for process in ranks:
        process.start()
    for process in ranks:
        process.join(180)
        assert process.exitcode == 0
    for rank in range(2):
        outcomes = open(f"{out}.{rank}").read().splitlines()
        assert len(outcomes) == 2 and all("Error" in outcome for outcome in outcomes), outcomes


_STUB_RCCL = r"""
// Test double of librccl.so for the CPU container: enough of the API for RcclCommunicator.__init__ to run.
// ncclCommInitRank records that it was entered (a real one would block waiting for the missing rank).
#include <stdio.h>
#include <string.h>
typedef struct { char internal[128]; } ncclUniqueId;
const char *ncclGetErrorString(int status) { return status ? "stub error" : "ok"; }
int ncclGetUniqueId(ncclUniqueId *id) { memset(id, 7, sizeof(*id)); return 0; }
int ncclCommInitRank(void **comm, int world, ncclUniqueId id, int rank) {
    FILE *mark = fopen(STUB_MARK, "a"); if (mark) { fprintf(mark, "entered %d\n", rank); fclose(mark); }
    *comm = (void *)1; return 0; }
int ncclAllGather(const void *s, void *r, size_t n, int t, void *c, void *st) { return 0; }
int ncclCommDestroy(void *c) { return 0; }
"""


def _rccl_one_rank_fails_worker(rank, world, port, out_path, stub, missing):
    sys.path.insert(0, ROOT)
    os.environ["DS_RCCL_LIBRARY"] = missing if rank == 1 else stub     # only rank 1 cannot load its library
    from doppel_speller_amd.distributed import RcclCommunicator, Rendezvous
    rendezvous = Rendezvous(rank, world, port=port)
    try:
        RcclCommunicator(rendezvous, rank, select_device=False)        # no GPU in this container
        outcome = "initialised"
    except Exception as error:  # noqa: BLE001 - the message is the result
        outcome = f"{type(error).__name__}: {error}"
    flags = rendezvous.all_gather_bytes(outcome.encode())
    with open(f"{out_path}.{rank}", "w") as handle:
        handle.write("\n".join(flag.decode() for flag in flags))
    rendezvous.close()


def test_rccl_failure_of_a_single_non_zero_rank_raises_everywhere(tmp_path):
    """rank 1 fails locally before ncclCommInitRank (its librccl cannot be loaded) while ranks 0 and 2
    are fine.  Every rank must raise after the readiness exchange and NOBODY may enter the collective ncclCommInitRank
    (it would wait for rank 1 forever).  librccl is a test double here: the container has no GPU."""
    import subprocess
    mark = str(tmp_path / "entered.txt")
    source = tmp_path / "stub_rccl.c"
    source.write_text(_STUB_RCCL)
    stub = str(tmp_path / "libstub_rccl.so")
    subprocess.check_call(["gcc", "-shared", "-fPIC", f'-DSTUB_MARK="{mark}"', str(source), "-o", stub])
    context = multiprocessing.get_context("spawn")
    port = _free_port()
    out = str(tmp_path / "outcome")
    ranks = [context.Process(target=_rccl_one_rank_fails_worker, args=(r, 3, port, out, stub, str(tmp_path / "no.so")))
             for r in range(3)]
    for process in ranks:
        process.start()
    for process in ranks:
        process.join(120)
        assert process.exitcode == 0
    for rank in range(3):
        outcomes = open(f"{out}.{rank}").read().splitlines()
        assert len(outcomes) == 3 and all("DoppelError" in outcome for outcome in outcomes), outcomes
        assert "another rank failed" in outcomes[0] and "rank 1" in outcomes[0] and "DS_RCCL_LIBRARY" in outcomes[1]
    assert not os.path.exists(mark)          # ncclCommInitRank was never entered

    # the same three ranks with a loadable library everywhere do enter it (the test double works)
    port = _free_port()
    ranks = [context.Process(target=_rccl_one_rank_fails_worker, args=(r, 3, port, out, stub, stub)) for r in range(3)]
    for process in ranks:
        process.start()
    for process in ranks:
        process.join(120)
        assert process.exitcode == 0
    assert open(f"{out}.0").read().splitlines() == ["initialised"] * 3
    assert sorted(open(mark).read().split()) == sorted("entered 0 entered 1 entered 2".split())


def test_rendezvous_ignores_a_silent_stray_connection(tmp_path):
    """A connection that sends nothing holds rank 0's accept loop for `hello_timeout`, not for the collective timeout."""
    import threading
    import time
    from doppel_speller_amd.distributed import Rendezvous
    port = _free_port()
    results = {}

    def rank0():
        results[0] = Rendezvous(0, 2, port=port, timeout=120.0, hello_timeout=1.0)

    server = threading.Thread(target=rank0)
    server.start()
    time.sleep(0.3)
    stray = socket.create_connection(("127.0.0.1", port))      # says nothing
    started = time.time()
    peer = Rendezvous(1, 2, port=port, timeout=120.0)
    server.join(30)
    assert 0 in results and time.time() - started < 20
    exchange = threading.Thread(target=lambda: results.__setitem__("a", results[0].all_gather_bytes(b"zero")))
    exchange.start()
    assert peer.all_gather_bytes(b"one") == [b"zero", b"one"]
    exchange.join(10)
    assert results["a"] == [b"zero", b"one"]
    stray.close()
    peer.close()
    results[0].close()


def test_published_workload_round_trip(tmp_path):
    """bench.py with N > 1: rank 0 publishes the whole workload as .npy files, the other ranks map it read-only and take
    their query shard -- every array identical, and the mapped form feeds the same host entry points."""
    from doppel_speller_amd import synth
    from doppel_speller_amd.distributed import shard_range, slice_queries
    w = synth.make_workload(4000, 301, seed=21)
    synth.publish_workload(w, str(tmp_path / "shared"))
    mapped = synth.load_workload(str(tmp_path / "shared"))
    assert (mapped.n_truth, mapped.n_queries, mapped.n_columns) == (4000, 301, w.n_columns)
    for name in synth._PUBLISHED:
        assert np.array_equal(np.asarray(getattr(mapped, name)), getattr(w, name)), name
        assert not getattr(mapped, name).flags.writeable
    begin, end = shard_range(301, 1, 2)
    a, b = slice_queries(w.q_rowptr, w.q_cols, w.q_maxint, begin, end), \
        slice_queries(mapped.q_rowptr, mapped.q_cols, mapped.q_maxint, begin, end)
    assert all(np.array_equal(x, y) for x, y in zip(a, b))
    assert synth.algorithmic_bytes_jaccard(mapped, 10, begin, end) + synth.algorithmic_bytes_jaccard(mapped, 10, 0, begin) == \
        synth.algorithmic_bytes_jaccard(w, 10)


def _published_worker(rank, world, port, shared, out_path):
    """One rank of bench.py's N > 1 flow on the CPU: rank 0 generates and publishes the workload, every rank maps it, takes
    its (uneven) query shard, answers it with the oracle and all-gathers the rows."""
    sys.path.insert(0, ROOT)
    from doppel_speller_amd import synth
    from doppel_speller_amd.distributed import HostCommunicator, Rendezvous, RowGather, shard_range, slice_queries
    from oracle import oracle
    rendezvous = Rendezvous(rank, world, port=port)
    if rank == 0:
        synth.publish_workload(synth.make_workload(2500, 1003, seed=23), shared)
    rendezvous.barrier()
    w = synth.load_workload(shared)
    begin, end = shard_range(w.n_queries, rank, world)
    rowptr, cols, maxint = slice_queries(w.q_rowptr, w.q_cols, w.q_maxint, begin, end)
    local = oracle.jaccard_topk(w.rowptr, w.truth_idx, w.idf32, w.sums32, rowptr, cols, maxint, 7)
    gathered = RowGather(HostCommunicator(rendezvous), w.n_queries, 7).gather(local)
    sizes = rendezvous.all_gather_bytes(str(end - begin).encode())
    rendezvous.barrier()
    np.save(f"{out_path}.{rank}.npy", gathered)
    with open(f"{out_path}.{rank}.sizes", "w") as handle:
        handle.write(" ".join(size.decode() for size in sizes))
    rendezvous.close()


def test_eight_ranks_share_one_published_workload_with_uneven_shards(tmp_path):
    """World size 8 -- the node BASELINE.json's C4 / C5 name -- end to end on the CPU: one published workload, 1003 queries
    over 8 ranks (shards of 126 and 125), the all-gather of every rank == one process over all the queries."""
    from doppel_speller_amd import synth
    from oracle import oracle
    oracle.build()
    world, port = 8, _free_port()
    shared, result = str(tmp_path / "shared"), str(tmp_path / "rows")
    context = multiprocessing.get_context("spawn")
    ranks = [context.Process(target=_published_worker, args=(r, world, port, shared, result)) for r in range(world)]
    for process in ranks:
        process.start()
    for process in ranks:
        process.join(300)
        assert process.exitcode == 0
    w = synth.make_workload(2500, 1003, seed=23)
    expected = oracle.jaccard_topk(w.rowptr, w.truth_idx, w.idf32, w.sums32, w.q_rowptr, w.q_cols, w.q_maxint, 7)
    for rank in range(world):
        assert np.array_equal(np.load(f"{result}.{rank}.npy"), expected)
        with open(f"{result}.{rank}.sizes") as handle:
            sizes = [int(x) for x in handle.read().split()]
        assert sum(sizes) == 1003 and sorted(set(sizes)) == [125, 126]
