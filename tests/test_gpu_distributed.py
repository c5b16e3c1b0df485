"""GPU-box rehearsal of the multi-GPU plumbing with the one GPU a test box has: the RCCL communicator (ctypes binding
of librccl, unique id through the rendezvous, ONE ncclAllGather on a HIP stream) with a single rank, and bench.py's
own rank launcher with the host communicator.  Real N > 1 RCCL runs need N GPUs: bench.py --gpus N."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_rccl_single_rank_gather_on_device(oracle):
    import ctypes
    import doppel_speller_amd as ds
    from doppel_speller_amd import _lib, synth
    from doppel_speller_amd.distributed import RcclCommunicator, Rendezvous, RowGather
    w = synth.make_workload(30000, 257, seed=19)
    pipeline = ds.CandidatePipeline(w, 10)
    stream = ctypes.c_void_p()
    _lib.check(_lib.lib().ds_stream_create(0, ctypes.byref(stream)), "stream")
    communicator = RcclCommunicator(Rendezvous(0, 1), 0)
    gather = RowGather(communicator, 257, 10)
    pipeline.enqueue_top_k(stream.value)
    gathered = gather.gather(pipeline.rows_ptr, stream.value)   # the ncclAllGather follows the kernels on the stream
    _lib.check(_lib.lib().ds_stream_sync(stream, 0), "sync")
    expected = oracle.jaccard_topk(w.rowptr, w.truth_idx, w.idf32, w.sums32, w.q_rowptr, w.q_cols, w.q_maxint, 10)
    assert np.array_equal(gathered.to_host(), expected)
    communicator.close()
    _lib.check(_lib.lib().ds_stream_destroy(stream, 0), "stream")


def test_bench_with_rccl_prints_one_json_line():
    """One rank, RCCL communicator (DS_BENCH_FORCE_DIST=1): stdout carries the JSON line and nothing else -- RCCL's
    version banner goes to stderr."""
    env = dict(os.environ, DS_BENCH_FORCE_DIST="1")
    env.pop("WORLD_SIZE", None)
    result = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--queries", "3000", "--truth", "60000",
                             "--k", "10", "--steps", "1", "--warmup", "1", "--cpu-seconds", "0", "--check", "32"],
                            env=env, capture_output=True, text=True, timeout=600)
    assert result.returncode == 0, result.stderr[-3000:]
    lines = [line for line in result.stdout.splitlines() if line.strip()]
    assert len(lines) == 1, result.stdout[:2000]
    line = json.loads(lines[0])
    assert line["rccl_ranks"] == 1 and line["verified_queries"] == 32 and "ncclAllGather" in line["collective"]


def test_bench_spawns_its_own_ranks():
    """`bench.py --gpus 2` without a launcher starts two fresh rank processes (both on the box's only GPU here, hence
    the host communicator) and rank 0 prints one JSON line for the whole job."""
    env = dict(os.environ)
    env.pop("WORLD_SIZE", None)
    env["DS_BENCH_SAME_DEVICE"] = "1"
    result = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--queries", "2000",
                             "--truth", "60000", "--k", "10", "--steps", "1", "--warmup", "1", "--cpu-seconds", "0",
                             "--check", "32", "--host-communicator"], env=env, capture_output=True, text=True,
                            timeout=600)
    assert result.returncode == 0, result.stderr[-3000:]
    line = json.loads(result.stdout.strip().splitlines()[-1])
    assert line["n_gpus"] == 2 and line["config"]["queries_per_gpu"] == 2000 and line["verified_queries"] == 32
    assert line["value"] > 0 and line["roofline"]["frac"] <= 1.0


def test_bench_under_the_driver_launcher_falls_back_loudly_when_rccl_refuses():
    """The driver's own command line (`python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr
    127.0.0.1 --master-port P bench.py --gpus 2 ...`) with both ranks on the box's only GPU: the launcher's store owns
    MASTER_PORT, the rendezvous finds its own port, RCCL refuses the duplicate device on both ranks, and the ranks agree
    to gather through the host -- one JSON line, `rccl_ranks` 0 and the reason on the line."""
    import socket
    with socket.socket() as probe:
        probe.bind(("127.0.0.1", 0))
        port = probe.getsockname()[1]
    env = dict(os.environ, DS_BENCH_SAME_DEVICE="1")
    for name in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR"):
        env.pop(name, None)
    result = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                             "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "bench.py"),
                             "--gpus", "2", "--queries", "2000", "--truth", "60000", "--k", "10", "--steps", "1",
                             "--warmup", "1", "--cpu-seconds", "0", "--check", "32"],
                            env=env, capture_output=True, text=True, timeout=900)
    assert result.returncode == 0, result.stderr[-3000:]
    lines = [line for line in result.stdout.splitlines() if line.startswith("{")]
    assert len(lines) == 1, result.stdout[:2000]
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["verified_queries"] == 32 and line["value"] > 0
    assert line["rccl_ranks"] == 0 and "ncclCommInitRank" in line["communicator_note"]


def _driver_launcher(arguments, env):
    import socket
    with socket.socket() as probe:
        probe.bind(("127.0.0.1", 0))
        port = probe.getsockname()[1]
    for name in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR"):
        env.pop(name, None)
    return subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "bench.py"),
                           "--gpus", "2"] + arguments, env=env, capture_output=True, text=True, timeout=900)


def test_bench_on_distinct_devices_exits_non_zero_when_rccl_cannot_start():
    """`--gpus 2` on a box with ONE GPU and no rehearsal switch: rank 1's device does not exist, so the communicator cannot
    be created.  Every rank learns it through the readiness exchange, nobody blocks in ncclCommInitRank, no JSON line is
    printed and the job's exit status is non-zero -- a scaling run never silently measures a TCP gather."""
    result = _driver_launcher(["--queries", "2000", "--truth", "60000", "--k", "10", "--steps", "1", "--warmup", "1",
                               "--cpu-seconds", "0", "--check", "8"], dict(os.environ))
    assert result.returncode != 0
    assert not [line for line in result.stdout.splitlines() if line.startswith("{")], result.stdout[:2000]
    assert "RCCL could not be initialised" in result.stderr


@pytest.mark.parametrize("config,k", [("C5", 100), ("C4", 50)])
def test_strong_scaling_rehearsal_two_ranks_share_one_published_workload(config, k):
    """C4's / C5's flow with two rank processes on the box's one GPU: rank 0 generates the WHOLE workload once (5M truth
    titles, top-50 / top-100 + features) and publishes it in shared memory, both ranks map it, build the replicated
    index and take UNEVEN query shards (10000 / 10001: a fixed number of queries in all, strong scaling); the rows are
    gathered through the host communicator."""
    env = dict(os.environ, DS_BENCH_SAME_DEVICE="1")
    env.pop("WORLD_SIZE", None)
    result = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--config", config, "--truth",
                             "5000000", "--queries", "20001", "--steps", "1", "--warmup", "1", "--cpu-seconds", "0",
                             "--check", "16", "--host-communicator"], env=env, capture_output=True, text=True,
                            timeout=1200)
    assert result.returncode == 0, result.stderr[-3000:]
    line = json.loads(result.stdout.strip().splitlines()[-1])
    assert line["n_gpus"] == 2 and line["scaling"] == "strong" and line["config"]["queries"] == 20001
    assert line["config"]["queries_per_gpu"] == 10000 and line["config"]["k"] == k and line["verified_queries"] == 16
    assert line["roofline"]["geometry"] == "narrow" and line["value"] > 0     # 5M rows: narrow tiles (wide above 10M)
    _assert_nothing_published_is_left()


def _assert_nothing_published_is_left():
    """rank 0 published under <shm>/ds_<uid>/bench_<port>_<random>/ (a 0700 directory of this user) and removed it."""
    private = os.path.join("/dev/shm", f"ds_{os.getuid()}")
    assert not os.path.isdir(private) or not [name for name in os.listdir(private) if name.startswith("bench_")]
    assert not [name for name in os.listdir("/dev/shm") if name.startswith("ds_bench_")]


def test_c5_rehearsal_two_ranks_at_the_real_50m_truth_rows():
    """the first 8-rank C5 run must not be the first time that concurrent 50M-row index builds share a host.
    Two rank processes on the box's one GPU (DS_BENCH_SAME_DEVICE=1): rank 0 generates and publishes the REAL 50M-row truth
    side once, both ranks map it, each builds its own replicated index (`cores / 2` host threads each, 2 x 20 GB on the
    device) and answers its uneven shard of 20,001 queries at top-100 + features.  Asserted: wall clock under 300 s, the
    answers verified, every rank's peak host RSS on the line (DESIGN.md section 7 states what to expect per rank)."""
    import time
    env = dict(os.environ, DS_BENCH_SAME_DEVICE="1")
    env.pop("WORLD_SIZE", None)
    started = time.perf_counter()
    result = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--config", "C5", "--queries",
                             "20001", "--steps", "1", "--warmup", "1", "--cpu-seconds", "0", "--check", "8",
                             "--host-communicator"], env=env, capture_output=True, text=True, timeout=900)
    wall = time.perf_counter() - started
    assert result.returncode == 0, result.stderr[-3000:]
    line = json.loads(result.stdout.strip().splitlines()[-1])
    print(f"C5 two-rank rehearsal at 50M rows: wall {wall:.0f} s, peak host RSS per rank {line['host_peak_rss_gib']} GiB, "
          f"published workload {line['published_workload_gib']:.1f} GiB")
    assert wall < 300.0, wall
    assert line["n_gpus"] == 2 and line["config"]["truth_titles"] == 50_000_000 and line["config"]["k"] == 100
    assert line["config"]["queries_per_gpu"] == 10000 and line["verified_queries"] == 8
    assert line["roofline"]["geometry"] == "wide" and len(line["host_peak_rss_gib"]) == 2
    assert max(line["host_peak_rss_gib"]) < 120.0, line["host_peak_rss_gib"]
    _assert_nothing_published_is_left()


def test_row_gather_pads_and_compacts_uneven_shards_on_the_device():
    """RowGather's device path with shards that differ by one row (C4 / C5 on a GPU count that does not divide the
    queries): the local rows are padded on the device, travel through ONE all_gather, and are compacted with
    device-to-device copies.  The communicator is a test double of RcclCommunicator (one process, two ranks' buffers):
    real RCCL needs two GPUs; everything else is the product path."""
    import ctypes
    from doppel_speller_amd import _lib
    from doppel_speller_amd.distributed import RowGather, shard_sizes
    n_queries, k = 2001, 7
    sizes = shard_sizes(n_queries, 2)
    rng = np.random.RandomState(3)
    rows = rng.randint(0, 1 << 30, (n_queries, k)).astype(np.int32)
    local = [_lib.DeviceArray.from_host(rows[:sizes[0]]), _lib.DeviceArray.from_host(rows[sizes[0]:])]
    padded_other = {}

    class TwoRanksInOneProcess:
        on_device = True
        world_size = 2

        def __init__(self, rank):
            self.rank = rank

        def all_gather(self, send, stream=None, out=None):
            row_bytes = send.shape[1] * 4
            mine = ctypes.c_void_p(out.ptr.value + self.rank * send.shape[0] * row_bytes)
            other = ctypes.c_void_p(out.ptr.value + (1 - self.rank) * send.shape[0] * row_bytes)
            _lib.check(_lib.lib().ds_memcpy_d2d_async(mine, send.ptr, send.shape[0] * row_bytes, 0, None), "copy")
            _lib.check(_lib.lib().ds_memcpy_d2d_async(other, padded_other[self.rank].ptr, send.shape[0] * row_bytes, 0,
                                                      None), "copy")
            return out

    longest = max(sizes)
    for rank in (0, 1):     # what the OTHER rank would send: its rows padded to the longest shard
        padded = np.full((longest, k), -1, dtype=np.int32)
        padded[:sizes[1 - rank]] = rows[:sizes[0]] if rank == 1 else rows[sizes[0]:]
        padded_other[rank] = _lib.DeviceArray.from_host(padded)
    for rank in (0, 1):
        gather = RowGather(TwoRanksInOneProcess(rank), n_queries, k)
        assert not gather.even and gather.local == sizes[rank]
        for _ in range(2):      # buffers are reused
            gathered = gather.gather(local[rank])
        _lib.check(_lib.lib().ds_stream_sync(None, 0), "sync")
        assert gathered.shape == (n_queries, k) and np.array_equal(gathered.to_host(), rows)
