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
                per-posting info, sums32 of dense scans, list pointers) / its average launch duration (HIP events
                on the launch stream); `traffic` = 2 * FETCH_SIZE + WRITE_SIZE of profiles/pmc_latest.json when that
                file was measured on this build and workload; `bound_model` = the kernel's shares of VALU issue, LDS and HBM time from the same
                counters; `speedup_over_reference_hbm_floor` = the bytes the REFERENCE's algorithm would read
                 / launch duration / 8 TB/s -- above 1 because the kernel skips most of them.
  cpu_baseline  the oracle (C restatement of the reference, OpenMP) timed on a bounded sample of the same workload
                (rank 0, N=1 only), best of a thread sweep.
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

    # thread sweep on small pilots (the per-thread N-vectors of the reference's algorithm thrash the caches when every
    # hardware thread runs one): 1 thread, then 32 / 64 / ... / all
    sweep = {}
    candidates = sorted({1, cores} | {t for t in (8, 16, 32, 64, 128) if t < cores})
    for threads in candidates:
        pilot = min(workload.n_queries, max(4 * threads, 128))  # enough queries per thread for a stable ranking
        tj, tf = run(pilot, threads)
        sweep[threads] = pilot * k / (tj + tf)
        if tj + tf > budget_seconds / 3:
            break
    best = max(sweep, key=sweep.get)
    n_sample = int(min(workload.n_queries, max(2 * best, 0.5 * budget_seconds * sweep[best] / k)))
    tj, tf = run(n_sample, best)
    oracle.set_num_threads(cores)
    return {"value": n_sample * k / (tj + tf), "unit": "candidate-pairs/s", "cores": best, "kind": "port",
            "sample": f"first {n_sample} queries of the workload x {workload.n_truth} truth titles, top-{k} "
                      f"(jaccard+topk {tj:.2f}s, features {tf:.2f}s on {best} OpenMP threads)",
            "cpu": cpu_model(), "host_threads": cores,
            "thread_sweep_pairs_per_s": {str(t): round(v, 1) for t, v in sweep.items()},
            "queries_per_s": n_sample / tj, "feature_pairs_per_s": n_sample * k / tf}


def resolve_config(name, world, queries=None, truth=None, k=None):
    """(queries per GPU, truth titles, k, "weak" | "strong", BASELINE.json's text, overridden?) of a configuration on
    `world` GPUs.  "per_gpu" configurations keep the per-GPU batch fixed (weak scaling); "total" ones divide a fixed
    number of queries over the GPUs (strong scaling: C4 = 8M in all, C5 = 1M in all)."""
    total_or_batch, mode, config_truth, config_k, text = CONFIGS[name]
    per_gpu = total_or_batch if mode == "per_gpu" else total_or_batch // world
    scaling = "weak" if mode == "per_gpu" else "strong"
    custom = queries is not None or truth is not None or k is not None
    if queries is not None:
        per_gpu, scaling = queries, "weak"
    return (per_gpu, truth if truth is not None else config_truth, k if k is not None else config_k, scaling, text,
            custom)


def spawn_ranks(gpus):
    """`--gpus N` without a launcher: start N fresh rank processes (one per GPU) BEFORE this process touches the GPU and
    exit with their worst status.  Rank 0 inherits stdout (the JSON line); the other ranks' stdout goes to stderr."""
    import socket
    with socket.socket() as probe:
        probe.bind(("127.0.0.1", 0))
        port = probe.getsockname()[1]
    ranks = []
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
    args = parser.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(args.gpus))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        log(f"error: --gpus {args.gpus} but WORLD_SIZE={world}")
        sys.exit(2)

    per_gpu, truth, k, scaling, text, custom = resolve_config(args.config, world, args.queries, args.truth, args.k)
    # DS_BENCH_FORCE_DIST=1 exercises the rendezvous / RCCL plumbing with a single rank (1-GPU rehearsal)
    distributed = world > 1 or os.environ.get("DS_BENCH_FORCE_DIST") == "1"

    import doppel_speller_amd as ds
    from doppel_speller_amd import _lib, synth
    from doppel_speller_amd.distributed import HostCommunicator, RcclCommunicator, Rendezvous, RowGather

    device = 0 if os.environ.get("DS_BENCH_SAME_DEVICE") == "1" else local_rank   # rehearsals on a one-GPU box
    rendezvous = communicator = gather = None
    if distributed:
        rendezvous = Rendezvous.from_environment()
        communicator_note = None
        if args.host_communicator:
            communicator = HostCommunicator(rendezvous)
        else:
            # The path itself has no collective (queries are sharded, the truth index is replicated): the gather only
            # assembles the result table.  If RCCL cannot be initialised the ranks AGREE (one flag each through the
            # rendezvous) to gather through the host instead, and the line says so (`rccl_ranks` 0 + `communicator_note`).
            try:
                communicator, failure = RcclCommunicator(rendezvous, device), b""
            except Exception as error:  # noqa: BLE001 - reported on the line, never hidden
                communicator, failure = None, f"rank {rank}: {error}".encode()[:400]
            failures = [f.decode() for f in rendezvous.all_gather_bytes(failure) if f]
            if failures:
                if communicator is not None:
                    communicator.close()
                communicator = HostCommunicator(rendezvous)
                communicator_note = "RCCL initialisation failed, rows gathered through the host: " + "; ".join(failures)
                log("warning: " + communicator_note)

    # ---- synthetic workload: truth replicated (same seed), queries distinct per rank
    t0 = time.perf_counter()
    stop = heartbeat("generating the workload") if rank == 0 else (lambda: None)
    workload = synth.make_workload(truth, per_gpu, seed=args.seed, query_seed=args.seed + 1000 * (rank + 1))
    stop()
    if rank == 0:
        log(f"workload: {time.perf_counter() - t0:.1f}s  {synth.workload_statistics(workload)}")
    t0 = time.perf_counter()
    stop = heartbeat("index build + upload") if rank == 0 else (lambda: None)
    pipeline = ds.CandidatePipeline(workload, k, device=device)
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
        gather = RowGather(communicator, per_gpu * world, k, device)

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

    # ---- spot check against the oracle (outside the timed region)
    checked = 0
    cells_per_pair = None
    if args.check > 0:
        from oracle import oracle
        n_check = min(args.check, per_gpu)
        rows = pipeline.rows()[:n_check]
        last = int(workload.q_rowptr[n_check])
        expected = oracle.jaccard_topk(workload.rowptr, workload.truth_idx, workload.idf32, workload.sums32,
                                       workload.q_rowptr[:n_check + 1], workload.q_cols[:last],
                                       workload.q_maxint[:n_check], k)
        assert np.array_equal(rows, expected), "top-k rows differ from the oracle"
        features = pipeline.features(n_check * k)
        pair_q = np.repeat(np.arange(n_check), k)
        pair_t = rows.reshape(-1)
        reference = oracle.construct_features(workload.q_len[pair_q], workload.t_len[pair_t], workload.q_enc[pair_q],
                                              workload.t_enc[pair_t], workload.t_counts[pair_t], 1, workload.n_truth)
        assert np.array_equal(features.view(np.uint32), reference.view(np.uint32)), "features differ from the oracle"
        checked = n_check
        # reference DP cells per pair (SURVEY 8d `cells(q,t)`) on the verified pairs: the unit of the features stage
        cells_per_pair = float(np.mean(oracle.feature_cells(workload.q_len[pair_q], workload.t_len[pair_t],
                                                            workload.q_enc[pair_q], workload.t_enc[pair_t], 1)))

    if rank == 0:
        pairs_per_step = per_gpu * k * world
        ms_per_step = 1000.0 * elapsed / args.steps
        reference_bytes = synth.algorithmic_bytes_jaccard(workload, k)
        mean_j = float(np.mean(jaccard_ms))
        mean_topk = float(np.mean(topk_kernel_ms))   # ds_jaccard_topk_kernel alone (the dominant kernel)
        requested = int(stats["requested_bytes"])     # bytes the kernel asked global memory for, last launch
        achieved = requested / (mean_topk * 1e-3) / 1e9
        roofline = {"bound": "hbm", "kernel": "ds_jaccard_topk_kernel", "achieved": achieved, "peak": HBM_PEAK_GBS,
                    "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": None,
                    "algorithmic_bytes_per_launch": requested, "avg_launch_ms": mean_topk,
                    "bytes_per_query": requested / max(1, per_gpu),
                    "note": "algorithmic bytes = what THIS kernel requests from global memory per launch (2-byte "
                            "postings of the traversed lists + per-posting row info, sums32 of the dense scans, list "
                            "pointers, per-column setup, refinement gathers, exact-stage probes), counted by a second "
                            "instantiation of the kernel in one extra untimed launch; `traffic` = 2 * FETCH_SIZE + WRITE_SIZE of profiles/pmc_latest.json "
                            "(FETCH_SIZE counts half of coalesced streams on gfx950) when measured on this build"}
        pmc_file = os.path.join(ROOT, "profiles", "pmc_latest.json")
        if os.path.exists(pmc_file):
            with open(pmc_file) as handle:
                pmc = json.load(handle)
            same = (pmc.get("queries") == per_gpu and pmc.get("truth") == truth and pmc.get("k") == k and
                    pmc.get("build_id") == _lib.lib().ds_build_id().decode())
            if same:  # counters of THIS build on THIS workload (scripts/profile_pmc.sh); stale files are ignored
                roofline["traffic"] = pmc.get("hbm_bytes_per_launch")
                roofline["bound_model"] = pmc.get("bound_model")
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
                                    f" [{per_gpu} queries per GPU x {truth} truth titles (replicated), tri-gram vocab "
                                    f"{workload.n_columns}, top-{k}" +
                                    (", one RCCL all-gather of the rows per step]" if world > 1 else "]")),
                       "name": args.config if not custom else "custom",
                       "queries_per_gpu": per_gpu, "truth_titles": truth, "k": k,
                       "seed": args.seed, "parallelism": f"query-shard x{world}"},
            "stages_ms": {"jaccard_topk": mean_j, "construct_features": float(np.mean(feature_ms)),
                          "ds_jaccard_topk_kernel": mean_topk,
                          "ds_jaccard_dense_kernel": float(np.mean(dense_kernel_ms))},
            "queries_per_s": per_gpu * world / (elapsed / args.steps),
            "per_stage": {
                "jaccard_queries_per_s": per_gpu / (mean_j * 1e-3),
                "jaccard_postings_per_s": (reference_bytes - per_gpu * (4 * truth + 4 * k)
                                           - 16 * int(workload.q_rowptr[-1])) / 4 / (mean_j * 1e-3),
                "feature_pairs_per_s": per_gpu * k / (float(np.mean(feature_ms)) * 1e-3),
                "feature_reference_dp_cells_per_s": None if cells_per_pair is None else
                    cells_per_pair * per_gpu * k / (float(np.mean(feature_ms)) * 1e-3),
                "reference_dp_cells_per_pair": cells_per_pair},
            "dense_path_queries": int(stats["dense_queries"]), "exact_candidates_per_query":
                stats["exact_candidates"] / max(1, per_gpu),
            "verified_queries": checked, "selections_per_query": stats["selections"] / max(1, per_gpu),
            "dense_reasons": stats["dense_reasons"],
            "tiles": {"sparse": stats["sparse_tiles"], "dense": stats["dense_tiles"]},
            "skipped_columns_per_query": stats["skipped_columns"] / max(1, per_gpu),
            "roofline": roofline,
            # SURVEY.md 8d's figure: what the REFERENCE's algorithm reads (4-byte postings of every query column, 4 N
            # bytes of sums per query ...).  The kernel skips most of it, hence a quotient above the HBM peak: this is
            # a speed-up over the reference algorithm's bandwidth floor, not an efficiency.
            "reference_algorithmic_bytes_per_launch": reference_bytes,
            "speedup_over_reference_hbm_floor": reference_bytes / (mean_topk * 1e-3) / 1e9 / HBM_PEAK_GBS,
            "build_id": _lib.lib().ds_build_id().decode(),
        }
        if distributed:
            line["rccl_ranks"] = world if communicator.on_device else 0
            if communicator_note:
                line["communicator_note"] = communicator_note
            line["collective"] = "ncclAllGather int32[queries_per_gpu, k] per step" if communicator.on_device else \
                "host all-gather through the TCP rendezvous (rehearsal)"
        if next_rows:
            line["next_rows"] = next_rows
        if any(stats["phase_cycles"].values()):  # library built with -DDS_DIAGNOSTICS and DS_PHASE_TIMERS=1
            line["diagnostics"] = {"phase_cycles": stats["phase_cycles"], "wave_refines": stats["refines"],
                                   "raw_entries_sparse": stats["raw_entries_sparse"],
                                   "bounds_record": stats.get("bounds_record")}
        if world == 1 and args.cpu_seconds > 0:
            line["cpu_baseline"] = cpu_baseline(workload, k, args.cpu_seconds)
        print(json.dumps(line), flush=True)
    if distributed:
        rendezvous.barrier()
        communicator.close()
        rendezvous.close()


if __name__ == "__main__":
    main()
