#!/usr/bin/env python3
"""Benchmark of the doppel-speller hot path on MI355X: candidate-pairs scored per second.

A step = one pass of the hot path over one batch of synthetic queries that is already resident in HBM:
Jaccard top-k of every query against the whole truth index, then construct_features on the Q*k surviving pairs
(metric of BASELINE.json: Q*k / (t_jaccard+topk + t_features)).

    python bench.py                                   # C2 on one GPU (BASELINE.json configs[1], the metric's config)
    python bench.py --config C3                       # 1M queries x 5M truth titles, top-50
    python bench.py --gpus 8 --config C4              # spawns 8 fresh rank processes itself (one per GPU)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \\
        bench.py --gpus N --steps K --warmup W        # the driver's launcher: RANK/LOCAL_RANK/WORLD_SIZE/MASTER_* are read

Configurations (BASELINE.json `configs`; per-GPU batch given, truth replicated on every GPU):
    C2  100k queries per GPU x 500k truth titles, top-10            (weak scaling with --gpus N)
    C3  1M queries per GPU x 5M truth titles, top-50                 (weak)
    C4  8M queries in all, sharded by query (1M per GPU at 8 GPUs) x 5M truth titles, top-50     (strong)
    C5  1M queries in all (125k per GPU at 8 GPUs) x 50M truth titles, top-100 + all features     (strong)

With N > 1 every rank (one process per GPU) owns its shard of the queries, the truth index is replicated, and each
step ends with the single all-gather of the int32 top-k rows over RCCL (ctypes binding, no PyTorch); barriers and
the max over ranks go through a TCP rendezvous.  Rank 0 prints ONE JSON line.

Extra objects on the line:
  roofline      ds_jaccard_topk_kernel against the HBM roofline: achieved = bytes the kernel REQUESTS from global
                memory per launch (counted by a second instantiation of the kernel in one untimed launch: postings,
                per-posting info, list pointers, row records of the refined rows) / its average launch duration (HIP events
                on the launch stream); `traffic` = 2 * FETCH_SIZE + WRITE_SIZE of profiles/pmc_latest.json when that
                file was measured on this build and workload; `bound_model` = the kernel's shares of VALU issue, LDS and HBM time from the same
                counters; `speedup_over_reference_hbm_floor` = the bytes the REFERENCE's algorithm would read
                 / launch duration / 8 TB/s -- above 1 because the kernel skips most of them.
  roofline_features  ds_construct_features_kernel against its own ceilings (VALU integer, not HBM).
  cpu_baseline  the oracle (C restatement of the reference, OpenMP) timed on a bounded sample of the same workload
                (rank 0, N=1 only), thread count chosen from pilots of >= 2 s each.

With N > 1 rank 0 generates the WHOLE workload once (all queries of all ranks: one vocabulary, as one MatchMaker over
all data has) and publishes it under /dev/shm; every rank maps it and takes its contiguous query shard.  RCCL is
mandatory then: if the communicator cannot be created the job prints the reason and exits non-zero (the host gather is
for rehearsals: --host-communicator, or --allow-host-fallback / DS_BENCH_SAME_DEVICE=1 on a one-GPU box).
"""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

HBM_PEAK_GBS = 8000.0     # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
SIMDS = 256 * 4           # 256 CUs x 4 SIMD-32: one VALU wave-instruction issues over 2 cycles (same guide)
CONFIGS = {
    # name: (queries, "per_gpu" | "total", truth titles, k, text of BASELINE.json's config)
    "C2": (100_000, "per_gpu", 500_000, 10, "100k synthetic queries x 500k truth titles, tri-gram vocab ~50k, top-10"),
    "C3": (1_000_000, "per_gpu", 5_000_000, 50, "1M queries x 5M truth titles, top-50"),
    "C4": (8_000_000, "total", 5_000_000, 50, "8M queries x 5M truth titles sharded by query, RCCL gather of top-k"),
    "C5": (1_000_000, "total", 50_000_000, 100, "1M queries x 50M truth titles, truth replicated per GPU, top-100 + "
                                               "full Levenshtein feature vector"),
}


def log(*args):
    print(*args, file=sys.stderr, flush=True)


def heartbeat(label, every=60.0):
    """A progress line on stderr every minute while a long host-side phase runs (the workload of C5 takes minutes to
    generate; a silent process looks hung to whoever watches it).  Returns a function that stops it."""
    import threading
    done = threading.Event()
    started = time.perf_counter()

    def beat():
        while not done.wait(every):
            log(f"... {label}: {time.perf_counter() - started:.0f}s")
    threading.Thread(target=beat, daemon=True).start()
    return done.set


def cpu_model():
    try:
        with open("/proc/cpuinfo") as handle:
            for line in handle:
                if line.startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(workload, k, budget_seconds):
    """Oracle on the host cores, bounded sample, best thread count of a sweep: the dict for the JSON line."""
    from oracle import oracle
    oracle.build()
    cores = oracle.num_threads()

    def run(n_sample, threads):
        oracle.set_num_threads(threads)
        first, last = int(workload.q_rowptr[0]), int(workload.q_rowptr[n_sample])
        t0 = time.perf_counter()
        rows = oracle.jaccard_topk(workload.rowptr, workload.truth_idx, workload.idf32, workload.sums32,
                                   workload.q_rowptr[:n_sample + 1], workload.q_cols[first:last],
                                   workload.q_maxint[:n_sample], k)
        t1 = time.perf_counter()
        pair_q = np.repeat(np.arange(n_sample), k)
        pair_t = rows.reshape(-1)
        oracle.construct_features(workload.q_len[pair_q], workload.t_len[pair_t], workload.q_enc[pair_q],
                                  workload.t_enc[pair_t], workload.t_counts[pair_t], 1, workload.n_truth)
        t2 = time.perf_counter()
        return t1 - t0, t2 - t1

    # thread sweep on pilots of >= 2 s each (long enough to leave the caches: the per-thread N-vectors of the
    # reference's algorithm thrash them when every hardware thread runs one): 1 thread, then 8 / 16 / ... / all
    sweep, pilots = {}, {}
    candidates = sorted({1, cores} | {t for t in (8, 16, 32, 64, 128) if t < cores})
    spent = 0.0
    for threads in candidates:
        pilot = min(workload.n_queries, max(4 * threads, 64))
        tj, tf = run(pilot, threads)
        while tj + tf < 2.0 and pilot < workload.n_queries:      # grow the pilot until it runs for two seconds
            pilot = min(workload.n_queries, int(pilot * max(2.0, 2.5 / max(tj + tf, 1e-3))))
            tj, tf = run(pilot, threads)
        sweep[threads] = pilot * k / (tj + tf)
        pilots[threads] = {"queries": pilot, "seconds": tj + tf}
        spent += tj + tf
        if spent > 0.6 * budget_seconds:
            break
    best = max(sweep, key=sweep.get)
    n_sample = int(min(workload.n_queries, max(pilots[best]["queries"], 0.5 * budget_seconds * sweep[best] / k)))
    tj, tf = run(n_sample, best)
    oracle.set_num_threads(cores)
    return {"value": n_sample * k / (tj + tf), "unit": "candidate-pairs/s", "cores": best, "kind": "port",
            "sample": f"first {n_sample} queries of the workload x {workload.n_truth} truth titles, top-{k} "
                      f"(jaccard+topk {tj:.2f}s, features {tf:.2f}s on {best} OpenMP threads)",
            "cpu": cpu_model(), "host_threads": cores,
            "pilot_same_threads": {"pairs_per_s": round(sweep[best], 1), **pilots[best]},
            "thread_sweep_pairs_per_s": {str(t): round(v, 1) for t, v in sweep.items()},
            "thread_counts_not_tried": [t for t in candidates if t not in sweep],   # the sweep stops when 60 % of the budget is spent
            "thread_sweep_note": "ascending thread counts until 60 % of --cpu-seconds is spent; counts listed in "
                                 "thread_counts_not_tried were NOT measured in this run (a longer --cpu-seconds measures them)",
            "thread_sweep_pilots": {str(t): v for t, v in pilots.items()},
            "queries_per_s": n_sample / tj, "feature_pairs_per_s": n_sample * k / tf}


def recurrence_trip_counts(q_len, t_len, q_enc, t_enc, space=1):
    """Text characters the features kernel's bit-parallel LCS recurrence steps through per pair (its dominant loop: one step =
    one character of the text against the 32- or 64-bit column vector), by call site, on a sample of pairs -- the trip counts
    that turn the static instruction counts of profiles/r04_features_isa_counts.txt into a dynamic picture.  Two pairs share a
    wave, so a wave steps through the LONGER of its two pairs' loops: `per_wave_pair` is that maximum, averaged.
    feature_engineering.py:106 lev(title, truth): text = the longer string; :128-149 word loop: per truth word (first 15),
    ceil(|title without spaces| / 32) batches of min(|word|, |title without spaces|) steps; :161-162 lev(reconstructed,
    truth): the reconstructed title is about as long as the truth title."""
    lq, lt = np.asarray(q_len, dtype=np.int64), np.asarray(t_len, dtype=np.int64)
    n = lq.shape[0]
    columns = np.arange(q_enc.shape[1])
    lw = ((np.asarray(q_enc) != space) & (columns[None, :] < lq[:, None])).sum(axis=1)
    word_steps = np.zeros(n, dtype=np.int64)
    short = np.zeros(n, dtype=np.int64)
    for i in range(n):
        text = np.asarray(t_enc[i, :lt[i]])
        cuts = np.concatenate(([-1], np.nonzero(text == space)[0], [lt[i]]))
        lengths = np.diff(cuts)[:15] - 1
        lengths = lengths[lengths > 0]
        if lw[i] > 0:
            word_steps[i] = int((-(-lw[i] // 32) * np.minimum(lengths, lw[i])).sum())
            short[i] = int((-(-lw[i] // 32) * np.minimum(lengths, lw[i]))[lengths <= 32].sum())
    whole = np.maximum(lq, lt)
    total = whole + word_steps + lt

    def per_wave(values):
        even = values[:n - n % 2].reshape(-1, 2).max(axis=1)
        return float(even.mean()) if even.shape[0] else float(values.mean())
    return {"title_vs_truth": float(whole.mean()), "word_loop": float(word_steps.mean()),
            "reconstructed_vs_truth": float(lt.mean()), "total": float(total.mean()), "per_wave_pair": per_wave(total),
            "share_on_the_32_bit_path": float((short.sum() + whole[np.minimum(lq, lt) <= 32].sum() +
                                               lt[np.minimum(lq, lt) <= 32].sum()) / max(1, total.sum()))}


def resolve_config(name, world, queries=None, truth=None, k=None):
    """(queries in the whole job, truth titles, k, "weak" | "strong", BASELINE.json's text, overridden?) of a
    configuration on `world` GPUs.  "per_gpu" configurations keep the per-GPU batch fixed (weak scaling); "total" ones
    divide a fixed number of queries over the GPUs (strong scaling: C4 = 8M in all, C5 = 1M in all).  Rank r owns the
    contiguous shard `shard_range(total, r, world)` (shards may differ by one query)."""
    total_or_batch, mode, config_truth, config_k, text = CONFIGS[name]
    total = total_or_batch * world if mode == "per_gpu" else total_or_batch
    scaling = "weak" if mode == "per_gpu" else "strong"
    custom = queries is not None or truth is not None or k is not None
    if queries is not None and mode == "per_gpu":
        total = queries * world            # --queries = per-GPU batch of a weak-scaling configuration
    elif queries is not None:
        total = queries                    # --queries = the job's total of a strong-scaling configuration
    return (total, truth if truth is not None else config_truth, k if k is not None else config_k, scaling, text,
            custom)


def spawn_ranks(gpus):
    """`--gpus N` without a launcher: start N fresh rank processes (one per GPU) BEFORE this process touches the GPU and
    exit with their worst status.  Rank 0 inherits stdout (the JSON line); the other ranks' stdout goes to stderr."""
    import socket
    from doppel_speller_amd import _lib, synth
    _lib.build_library()          # hipcc only, no GPU call: the ranks must not race to rebuild a stale library
    synth.build_native()
    with socket.socket() as probe:
        probe.bind(("127.0.0.1", 0))
        port = probe.getsockname()[1]
    ranks = []
    os.environ.setdefault("DS_HOST_THREADS", str(max(2, min(32, len(os.sched_getaffinity(0)) // gpus))))
    for rank in range(gpus):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(gpus), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        ranks.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=None if rank == 0 else sys.stderr))
    return max(process.wait() for process in ranks)


def main():
    parser = argparse.ArgumentParser()
    parser.add_argument("--gpus", type=int, default=1)
    parser.add_argument("--steps", type=int, default=3)
    parser.add_argument("--warmup", type=int, default=1)
    parser.add_argument("--config", choices=sorted(CONFIGS), default="C2")
    parser.add_argument("--queries", type=int, default=None, help="queries per GPU (overrides the configuration)")
    parser.add_argument("--truth", type=int, default=None)
    parser.add_argument("--k", type=int, default=None)
    parser.add_argument("--seed", type=int, default=20260101)
    parser.add_argument("--cpu-seconds", type=float, default=20.0, help="CPU baseline budget (0 disables)")
    parser.add_argument("--check", type=int, default=64, help="queries verified against the oracle after the run")
    parser.add_argument("--host-communicator", action="store_true",
                        help="gather through the TCP rendezvous instead of RCCL (rehearsals without N GPUs)")
    parser.add_argument("--allow-host-fallback", action="store_true",
                        help="if RCCL cannot be initialised, gather through the host instead of failing (rehearsals)")
    parser.add_argument("--shared-dir", default=None, help="where rank 0 publishes the workload (default /dev/shm)")
    args = parser.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(args.gpus))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        log(f"error: --gpus {args.gpus} but WORLD_SIZE={world}")
        sys.exit(2)

    total_queries, truth, k, scaling, text, custom = resolve_config(args.config, world, args.queries, args.truth, args.k)
    # DS_BENCH_FORCE_DIST=1 exercises the rendezvous / RCCL plumbing with a single rank (1-GPU rehearsal)
    distributed = world > 1 or os.environ.get("DS_BENCH_FORCE_DIST") == "1"

    import doppel_speller_amd as ds
    from doppel_speller_amd import _lib, synth
    from doppel_speller_amd.distributed import (HostCommunicator, RcclCommunicator, Rendezvous, RowGather,
                                                private_directory, shard_range)

    same_device = os.environ.get("DS_BENCH_SAME_DEVICE") == "1"     # rehearsals: every rank on a one-GPU box's GPU
    device = 0 if same_device else local_rank
    q_begin, q_end = shard_range(total_queries, rank, world)
    per_gpu = q_end - q_begin
    rendezvous = communicator = gather = None
    if distributed:
        rendezvous = Rendezvous.from_environment()
        communicator_note = None
        if args.host_communicator:
            communicator = HostCommunicator(rendezvous)
        else:
            # A job asked to run on N GPUs does not quietly measure a TCP gather: when RCCL cannot be initialised on any
            # rank, every rank learns it (one flag each through the rendezvous), prints the reason and exits non-zero --
            # unless a rehearsal switch allows the host gather, which the line then reports (`rccl_ranks` 0 + note).
            try:
                communicator, failure = RcclCommunicator(rendezvous, device), b""
            except Exception as error:  # noqa: BLE001 - reported by every rank, fatal unless a rehearsal switch is set
                communicator, failure = None, f"rank {rank}: {type(error).__name__}: {error}".encode()[:400]
            failures = [f.decode("utf-8", "replace") for f in rendezvous.all_gather_bytes(failure) if f]
            if failures:
                if communicator is not None:
                    communicator.close()
                if not (args.allow_host_fallback or same_device):
                    log(f"rank {rank}: RCCL could not be initialised ({'; '.join(failures)}); no line is printed.  "
                        "Rehearsals without N GPUs: --host-communicator or --allow-host-fallback")
                    rendezvous.close()
                    sys.exit(3)
                communicator = HostCommunicator(rendezvous)
                communicator_note = "RCCL initialisation failed, rows gathered through the host: " + "; ".join(failures)
                if rank == 0:
                    log("warning: " + communicator_note)

    # ---- synthetic workload.  One rank: generated in place.  Several ranks: rank 0 generates ALL of it (truth side and
    # the queries of every rank: one vocabulary, as one MatchMaker over all the data has), publishes it as .npy files in
    # shared memory, and every rank maps it and takes its shard -- not N times the minutes and the gigabytes.
    t0 = time.perf_counter()
    shared = None
    if world == 1:
        stop = heartbeat("generating the workload")
        workload = synth.make_workload(truth, total_queries, seed=args.seed)
        experiment = os.environ.get("DS_EXPERIMENT_QUERY_ORDER", "")  # tuning experiments only (profiles/r03_tuning.txt)
        if experiment:
            key = workload.q_maxint if "maxint" in experiment else np.diff(workload.q_rowptr)
            order = np.argsort(key, kind="stable")
            workload = synth.reorder_queries(workload, order[::-1] if experiment.endswith("desc") else order)
        stop()
    else:
        import shutil
        import tempfile
        failure = b""
        if rank == 0:
            stop = heartbeat("generating the workload")
            try:
                # published inside a 0700 directory of this user (verified: a real directory, owned, not a planted link), in a
                # fresh subdirectory whose name nobody can predict; its path travels through the rendezvous
                base = private_directory(args.shared_dir or ("/dev/shm" if os.path.isdir("/dev/shm") else None))
                shared = tempfile.mkdtemp(prefix=f"bench_{os.environ.get('MASTER_PORT', '0')}_", dir=base)
                import atexit
                atexit.register(shutil.rmtree, shared, True)
                synth.publish_workload(synth.make_workload(truth, total_queries, seed=args.seed), shared)
            except Exception as error:  # noqa: BLE001 - every rank must learn about it
                failure = f"{type(error).__name__}: {error}".encode()[:400]
            stop()
        published = rendezvous.broadcast_bytes((b"E" + failure if failure else b"P" + shared.encode()) if rank == 0 else None)
        if published[:1] != b"P":
            log(f"rank {rank}: rank 0 could not generate the workload: {published[1:].decode('utf-8', 'replace')}")
            sys.exit(4)
        shared = published[1:].decode()
        workload = synth.load_workload(shared)
    if rank == 0:
        log(f"workload: {time.perf_counter() - t0:.1f}s  {synth.workload_statistics(workload)}")
    t0 = time.perf_counter()
    stop = heartbeat("index build + upload") if rank == 0 else (lambda: None)
    pipeline = ds.CandidatePipeline(workload, k, device=device, q_begin=q_begin, q_end=q_end)
    stop()
    if rank == 0:
        log(f"upload + index build: {time.perf_counter() - t0:.1f}s  {pipeline.index.info()}")
    stream_handle = None
    stream = 0
    if distributed:
        import ctypes
        stream_handle = ctypes.c_void_p()
        _lib.check(_lib.lib().ds_stream_create(device, ctypes.byref(stream_handle)), "ds_stream_create")
        stream = stream_handle.value
        gather = RowGather(communicator, total_queries, k, device)

    def barrier():
        _lib.check(_lib.lib().ds_stream_sync(stream_handle, device), "sync")
        if distributed:
            rendezvous.barrier()

    timer_j, timer_f = _lib.Timer(device), _lib.Timer(device)
    jaccard_ms, feature_ms, topk_kernel_ms, dense_kernel_ms = [], [], [], []

    def step(record):
        timer_j.start(stream)
        pipeline.enqueue_top_k(stream)
        timer_j.stop(stream)
        if os.environ.get("DS_BENCH_SYNC_EACH") == "1":  # fault localisation: which stage was running
            log("top-k enqueued"); pipeline.sync(stream); log("top-k done")
        timer_f.start(stream)
        pipeline.enqueue_features(stream)
        timer_f.stop(stream)
        if os.environ.get("DS_BENCH_SYNC_EACH") == "1":
            _lib.check(_lib.lib().ds_stream_sync(stream_handle, device), "sync"); log("features done")
        if distributed:
            if communicator.on_device:
                gather.gather(pipeline.rows_ptr, stream)          # ONE ncclAllGather on the same stream
            else:
                gather.gather(pipeline.rows())
        if record:
            jaccard_ms.append(timer_j.elapsed_ms())   # synchronises on the stop events
            feature_ms.append(timer_f.elapsed_ms())
            kernel_times = pipeline.sync(stream)       # HIP events recorded by the library around each kernel
            topk_kernel_ms.append(kernel_times["topk_kernel_ms"])
            dense_kernel_ms.append(kernel_times["dense_kernel_ms"])

    for _ in range(args.warmup):
        step(False)
    barrier()
    t_begin = time.perf_counter()
    for _ in range(args.steps):
        step(True)
    barrier()
    elapsed = time.perf_counter() - t_begin
    stats = pipeline.sync(stream)
    if distributed:
        elapsed = rendezvous.max(elapsed)
    # one more launch, untimed, with the instantiation of the kernel that counts the bytes it requests (same work)
    pipeline.index.option("count_bytes", 1)
    pipeline.enqueue_top_k(stream)
    stats["requested_bytes"] = pipeline.sync(stream)["requested_bytes"]
    pipeline.index.option("count_bytes", 0)

    # ---- next rows on the same resident data, timed on their own (not part of the metric)
    next_rows = {}
    if not distributed:
        timer_c = _lib.Timer(device)
        pipeline.enqueue_close_matches(stream)          # first call allocates its outputs
        timer_c.start(stream)
        pipeline.enqueue_close_matches(stream)
        timer_c.stop(stream)
        next_rows["close_matches_ms"] = timer_c.elapsed_ms()
        next_rows["close_match_pairs_per_s"] = per_gpu * k / (next_rows["close_matches_ms"] * 1e-3)
        from doppel_speller_amd import ForestModel
        f = synth.make_forest()                              # 300 random trees of depth 6 (the model file is not in the tree)
        model = ForestModel(f["feature"], f["threshold"], f["yes"], f["no"], f["missing"], f["tree_offsets"],
                            f["n_features"], f["base_margin"], device)
        pipeline.enqueue_predict(model, stream)             # first call allocates its output
        timer_c.start(stream)
        pipeline.enqueue_predict(model, stream)
        timer_c.stop(stream)
        next_rows["forest_predict_ms"] = timer_c.elapsed_ms()
        next_rows["forest"] = "300 random trees, depth 6, 66 features"

    # ---- the reference's own call surface on the same data (outside the timed region, one GPU): one get_closest_matches
    # per row in a dict comprehension (predict.py:126-127), then ONE 9-argument construct_features over the pairs
    # (predict.py:215-219) -- host arrays in, host arrays out, PCIe included.  The gathers that build the 9 arguments
    # (predict.py:195-204) are the caller's and are not timed.  At most ~1M pairs (the padded arguments are 572 B per pair).
    surface = None
    if not distributed and os.environ.get("DS_BENCH_SURFACE", "1") != "0":
        from doppel_speller_amd.distributed import slice_queries
        n_surface = int(min(per_gpu, max(1, 1_000_000 // k)))
        s_rowptr, s_cols, s_maxint = slice_queries(workload.q_rowptr, workload.q_cols, workload.q_maxint, q_begin,
                                                   q_begin + n_surface)
        match_maker = ds.MatchMaker.from_index(pipeline.index, s_rowptr, s_cols, s_maxint, workload.title_id, k)
        t_a = time.perf_counter()
        nearest = {row: match_maker.get_closest_matches(row) for row in range(n_surface)}
        t_b = time.perf_counter()
        surface_rows = match_maker.get_closest_matches_batch()
        assert nearest[n_surface - 1] == workload.title_id[surface_rows[n_surface - 1]].tolist()
        pair_q = q_begin + np.repeat(np.arange(n_surface), k)
        pair_t = surface_rows.reshape(-1)
        arguments = (np.ascontiguousarray(workload.q_len[pair_q]), np.ascontiguousarray(workload.t_len[pair_t]),
                     np.ascontiguousarray(workload.q_enc[pair_q]), np.ascontiguousarray(workload.t_enc[pair_t]),
                     np.ascontiguousarray(workload.t_counts[pair_t]))
        surface_features = np.zeros((pair_q.shape[0], ds.FEATURES_COUNT), dtype=np.float32)
        dummy = np.zeros(ds.FEATURES_COUNT, dtype=np.uint8)
        feature_seconds = []
        for _ in range(2):   # the first call allocates the pinned staging slots (kept for the process's life)
            t_c = time.perf_counter()
            ds.construct_features(*arguments, ds.SPACE_CODE, workload.n_truth, dummy, surface_features)
            feature_seconds.append(time.perf_counter() - t_c)
        n_pairs = int(pair_q.shape[0])
        surface = {"queries": n_surface, "pairs": n_pairs,
                   "get_closest_matches_loop_s": t_b - t_a,
                   "get_closest_matches_us_per_call": 1e6 * (t_b - t_a) / n_surface,
                   "construct_features_9arg_first_call_s": feature_seconds[0],
                   "construct_features_9arg_s": feature_seconds[1],
                   "pairs_per_s": n_pairs / ((t_b - t_a) + feature_seconds[1]),
                   "pairs_per_s_first_call": n_pairs / ((t_b - t_a) + feature_seconds[0]),
                   "host_bytes_per_pair": 572 + 264,
                   "note": "host arrays in, host arrays out (PCIe and the host-side packing included): dict comprehension of "
                           "get_closest_matches over the rows (first call = one batched launch through the host-pointer entry "
                           "point + one title_id look-up for all rows) + one 9-argument construct_features over the pairs; "
                           "the caller's gathers that build the 9 arguments are not timed"}
        surface_check = surface_features
        del arguments, nearest
    else:
        surface_check = None

    # ---- spot check against the oracle (outside the timed region)
    checked = 0
    cells_per_pair = recurrence_steps = None
    if args.check > 0:
        from oracle import oracle
        n_check = min(args.check, per_gpu)
        rows = pipeline.rows()[:n_check]
        first, last = int(workload.q_rowptr[q_begin]), int(workload.q_rowptr[q_begin + n_check])
        expected = oracle.jaccard_topk(workload.rowptr, workload.truth_idx, workload.idf32, workload.sums32,
                                       np.asarray(workload.q_rowptr[q_begin:q_begin + n_check + 1]) - first,
                                       workload.q_cols[first:last], workload.q_maxint[q_begin:q_begin + n_check], k)
        assert np.array_equal(rows, expected), "top-k rows differ from the oracle"
        features = pipeline.features(n_check * k)
        pair_q = q_begin + np.repeat(np.arange(n_check), k)
        pair_t = rows.reshape(-1)
        reference = oracle.construct_features(workload.q_len[pair_q], workload.t_len[pair_t], workload.q_enc[pair_q],
                                              workload.t_enc[pair_t], workload.t_counts[pair_t], 1, workload.n_truth)
        assert np.array_equal(features.view(np.uint32), reference.view(np.uint32)), "features differ from the oracle"
        checked = n_check
        if surface_check is not None:   # the surface's features are the pipeline's (same pairs, other entry point)
            assert np.array_equal(surface_check[:n_check * k].view(np.uint32), reference.view(np.uint32)), \
                "9-argument construct_features differs from the oracle"
        # reference DP cells per pair (SURVEY 8d `cells(q,t)`) on the verified pairs: the unit of the features stage
        cells_per_pair = float(np.mean(oracle.feature_cells(workload.q_len[pair_q], workload.t_len[pair_t],
                                                            workload.q_enc[pair_q], workload.t_enc[pair_t], 1)))
        recurrence_steps = recurrence_trip_counts(workload.q_len[pair_q], workload.t_len[pair_t], workload.q_enc[pair_q],
                                                  workload.t_enc[pair_t])

    host_rss, published_bytes = None, 0
    if distributed:
        import resource
        import struct
        mine = resource.getrusage(resource.RUSAGE_SELF).ru_maxrss / 2 ** 20        # KiB -> GiB
        host_rss = [round(struct.unpack("<d", raw)[0], 2) for raw in rendezvous.all_gather_bytes(struct.pack("<d", mine))]
        if shared is not None and rank == 0:
            published_bytes = sum(os.path.getsize(os.path.join(shared, name)) for name in os.listdir(shared))
    if rank == 0:
        pairs_per_step = total_queries * k
        ms_per_step = 1000.0 * elapsed / args.steps
        reference_bytes = synth.algorithmic_bytes_jaccard(workload, k, q_begin, q_end)   # this rank's shard, like the kernel time
        shard_columns = int(workload.q_rowptr[q_end]) - int(workload.q_rowptr[q_begin])
        mean_j = float(np.mean(jaccard_ms))
        mean_f = float(np.mean(feature_ms))
        mean_topk = float(np.mean(topk_kernel_ms))   # ds_jaccard_topk_kernel alone (the dominant kernel)
        requested = int(stats["requested_bytes"])     # bytes the kernel asked global memory for, last launch
        achieved = requested / (mean_topk * 1e-3) / 1e9
        build_id = _lib.lib().ds_build_id().decode()
        roofline = {"bound": "hbm", "kernel": "ds_jaccard_topk_kernel", "achieved": achieved, "peak": HBM_PEAK_GBS,
                    "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": None,
                    "requested_bytes_per_launch": requested, "survey_8d_bytes_per_launch": reference_bytes,
                    "frac_on_survey_8d_bytes": reference_bytes / (mean_topk * 1e-3) / 1e9 / HBM_PEAK_GBS,
                    "avg_launch_ms": mean_topk, "geometry": "narrow" if pipeline.index.info()["tile_rows"] == 12288 else "wide",
                    "bytes_per_query": requested / max(1, per_gpu),
                    "note": "`achieved` / `frac` use requested_bytes_per_launch = what THIS kernel requests from global "
                            "memory per launch (2-byte postings of the traversed lists + per-posting row info, list pointers, "
                            "per-column setup, refinement gathers, exact-stage probes; dense scans read nothing), "
                            "counted by a second instantiation of the kernel in one extra untimed launch.  "
                            "survey_8d_bytes_per_launch = SURVEY.md 8d's B_jac, what the REFERENCE's algorithm reads: the "
                            "kernel skips most of it (MaxScore), so frac_on_survey_8d_bytes exceeds 1 -- a speed-up over "
                            "the reference algorithm's bandwidth floor, not an efficiency.  `traffic` = 2 * FETCH_SIZE + "
                            "WRITE_SIZE per launch from the entry of profiles/pmc_latest.json measured on this build "
                            "and workload (FETCH_SIZE counts half of coalesced streams on gfx950), else null.  `bound` = the roofline "
                            "this fraction is priced against (SURVEY.md 8d), `limited_by` = what the counters say limits the "
                            "kernel, `memory_level` = where the index lives (an index of at most 256 MiB is Infinity-Cache "
                            "resident: 8 TB/s of HBM is then not its memory ceiling)"}
        features_bound_model = None
        pmc_file = os.path.join(ROOT, "profiles", "pmc_latest.json")
        roofline["frac_on_counter_traffic"] = None
        roofline["limited_by"] = None
        if os.path.exists(pmc_file):
            with open(pmc_file) as handle:
                pmc = json.load(handle)
            for entry in pmc.get("entries", [pmc]):   # one entry per (workload, build id): scripts/profile_pmc.sh
                if (entry.get("queries") == per_gpu and entry.get("truth") == truth and entry.get("k") == k and
                        entry.get("build_id") == build_id):   # stale entries are ignored
                    roofline["traffic"] = entry.get("hbm_bytes_per_launch")
                    roofline["bound_model"] = entry.get("bound_model")
                    features_bound_model = entry.get("features_bound_model")
                    roofline["counters_source"] = ("profiles/pmc_latest.json: rocprofv3 PMC passes of this build and workload "
                                                   "(scripts/profile_pmc.sh), NOT measured in this run")
                    if roofline["traffic"]:
                        roofline["frac_on_counter_traffic"] = roofline["traffic"] / (mean_topk * 1e-3) / 1e9 / HBM_PEAK_GBS
                    if isinstance(roofline["bound_model"], dict):
                        roofline["limited_by"] = roofline["bound_model"].get("binding")
        # `bound` names the roofline the fraction is priced against (HBM); `limited_by` what the counter model
        # says holds the kernel (instruction issue: no memory level is near its peak); `memory_level` where the index lives.
        index_bytes = int(pipeline.index.info()["device_bytes"])
        roofline["index_bytes"] = index_bytes
        roofline["memory_level"] = "infinity cache (the index fits its 256 MiB)" if index_bytes <= 256 * 2 ** 20 else "hbm"
        # ds_construct_features_kernel: bound by VALU integer work, not HBM.  Ceiling of the REFERENCE's DP
        # formulation: 256 CUs x 128 lanes per clock (the f32 vector peak of the microarchitecture guide, 157.3 TF = 2 x
        # 256 x 128 x 2.4 GHz) / 5 integer operations per DP cell.  The kernel computes LCS bit-parallel (64 cells per 64-bit
        # step), so this is reference-cells per second against the cost of computing them literally.
        valu_cells_peak = 256 * 128 * 2.4e9 / 5.0
        cells_rate = None if cells_per_pair is None else cells_per_pair * per_gpu * k / (mean_f * 1e-3)
        feature_bytes = per_gpu * k * 330                       # indexed form: 8 B of indexes + 264 B out + ~58 B of titles
        roofline_features = {
            "bound": "valu-int", "kernel": "ds_construct_features_kernel", "achieved": cells_rate,
            "peak": valu_cells_peak, "unit": "reference DP cells/s",
            "frac": None if cells_rate is None else cells_rate / valu_cells_peak,
            "hbm_frac": feature_bytes / (mean_f * 1e-3) / 1e9 / HBM_PEAK_GBS, "hbm_bytes_per_pair": 330,
            "avg_launch_ms": mean_f, "reference_dp_cells_per_pair": cells_per_pair,
            "recurrence_steps_per_pair": recurrence_steps,
            "bound_model": features_bound_model,
            "note": "peak = 256 CUs x 128 integer lanes x 2.4 GHz / 5 operations per DP cell (SURVEY.md 8d); cells counted "
                    "by the oracle's feature_cells on the verified pairs"}
        line = {
            "metric": "candidate-pairs scored/sec (Jaccard top-k + Levenshtein), 100k x 500k titles",
            "value": pairs_per_step / (elapsed / args.steps),
            "unit": "candidate-pairs/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step,
            "higher_is_better": True, "scaling": scaling, "vs_baseline": None,
            "dtype": "f32 accumulate / f64 finalise (Jaccard), u8 + f64 ratio (Levenshtein)",
            "data": "synthetic",
            "config": {"workload": (f"{args.config}: " if not custom else "custom: ") +
                                   (f"{world}xMI355X: " + (text if not custom else "") +
                                    f" [{total_queries} queries in all, {per_gpu} per GPU x {truth} truth titles (replicated), tri-gram vocab "
                                    f"{workload.n_columns}, top-{k}" +
                                    (", one RCCL all-gather of the rows per step]" if world > 1 else "]")),
                       "name": args.config if not custom else "custom",
                       "queries_per_gpu": per_gpu, "queries": total_queries, "truth_titles": truth, "k": k,
                       "seed": args.seed, "parallelism": f"query-shard x{world}"},
            "stages_ms": {"jaccard_topk": mean_j, "construct_features": mean_f,
                          "ds_jaccard_topk_kernel": mean_topk,
                          "ds_jaccard_dense_kernel": float(np.mean(dense_kernel_ms))},
            "queries_per_s": total_queries / (elapsed / args.steps),
            "per_stage": {
                "jaccard_queries_per_s": per_gpu / (mean_j * 1e-3),
                "jaccard_postings_per_s": (reference_bytes - per_gpu * (4 * truth + 4 * k) - 16 * shard_columns) / 4 /
                                          (mean_j * 1e-3),
                "feature_pairs_per_s": per_gpu * k / (mean_f * 1e-3),
                "feature_reference_dp_cells_per_s": cells_rate,
                "reference_dp_cells_per_pair": cells_per_pair},
            "dense_path_queries": int(stats["dense_queries"]), "exact_candidates_per_query":
                stats["exact_candidates"] / max(1, per_gpu),
            "verified_queries": checked, "selections_per_query": stats["selections"] / max(1, per_gpu),
            "dense_reasons": stats["dense_reasons"], "sparse_redos": int(stats["sparse_redos"]),
            "bounds_record": [int(x) for x in stats["bounds_record"]],   # all 0 unless a -DDS_BOUNDS_CHECK build caught an index
            "tiles": {"sparse": stats["sparse_tiles"], "dense": stats["dense_tiles"]},
            "skipped_columns_per_query": stats["skipped_columns"] / max(1, per_gpu),
            "roofline": roofline,
            "roofline_features": roofline_features,
            "speedup_over_reference_hbm_floor": roofline["frac_on_survey_8d_bytes"],
            "gpu_seconds_in_timed_region": elapsed,
            "host_setup_note": "workload generation, index build and the CPU baseline run on the host before / after the "
                               "timed region; the GPU is busy for `gpu_seconds_in_timed_region` of the process's life",
            "build_id": build_id,
        }
        if distributed:
            line["host_peak_rss_gib"] = host_rss          # every rank's peak resident set (the mapped workload's pages included)
            line["published_workload_gib"] = published_bytes / 2 ** 30
            line["rccl_ranks"] = world if communicator.on_device else 0
            if communicator_note:
                line["communicator_note"] = communicator_note
            line["collective"] = "ncclAllGather int32[queries_per_gpu, k] per step" if communicator.on_device else \
                "host all-gather through the TCP rendezvous (rehearsal)"
        if surface is not None:
            line["surface"] = surface
        if next_rows:
            line["next_rows"] = next_rows
        if any(stats["phase_cycles"].values()):  # library built with -DDS_DIAGNOSTICS and DS_PHASE_TIMERS=1
            line["diagnostics"] = {"phase_cycles": stats["phase_cycles"], "wave_refines": stats["refines"],
                                   "raw_entries": stats["raw_entries"], "refine_survivors": stats["refine_survivors"],
                                   "raw_entries_sparse": stats["raw_entries_sparse"]}
        if world == 1 and args.cpu_seconds > 0:
            line["cpu_baseline"] = cpu_baseline(workload, k, args.cpu_seconds)
        print(json.dumps(line), flush=True)
    if distributed:
        rendezvous.barrier()
        communicator.close()
        rendezvous.close()
    if shared is not None and rank == 0:
        import shutil
        shutil.rmtree(shared, ignore_errors=True)


if __name__ == "__main__":
    main()
