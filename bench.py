#!/usr/bin/env python3
"""bench.py — Mutations-scan throughput of the MI355X filter engine, with roofline and CPU baseline.

Contract (driver): `python bench.py --gpus N --steps K --warmup W`; for N > 1 it is launched under
torch.distributed.run with one rank per GPU.  Rank 0 prints ONE JSON line.

A *step* is one complete query through the C++ QueryEngine mirror:
    {"action": {"type": "Mutations", "minProportion": 0.05},
     "filterExpression": {"type": "PangoLineage", "column": "pango_lineage", "value": "B.1", "includeSublineages": true}}
i.e. JSON parse -> filter compile -> lineage bitset (HBM resident) -> Mutations scan (plane rows, escape keys, runs of the
missing symbol, derived symbols) -> (N > 1: all-reduce of the count table over RCCL) -> row selection on the device, the
rows written into page-locked host memory -> response JSON.  The store is resident in HBM before the timed region.

Workload (default): the north-star target "10 M-sequence Mutations scan" — 10 M synthetic SARS-CoV-2 sequences x 29 903 nt
(BASELINE.json configs[3], nucleotide leg, with configs[1]'s PangoLineage filter) on ONE database that also holds the 12
genes.  N > 1 (`--shard position`, the north star's split): the same database sharded by position range, one all-reduce
of the count table per query (strong scaling); the line then also carries the amino-acid leg on those shards, and
(`--shard sequence`) BASELINE.json configs[4] — a 50 M-sequence database (genome + 12 genes) sharded by sequence id,
ONE batch of 100 filter + Mutations / AminoAcidMutations queries with an all-reduce per count table — and the filter
queries / s on those shards.  At N = 1 configs[4] runs whole on the one GPU.

metric = positions x sequences / s (whole job).  roofline: the dominant kernel of the scan (by HIP events on the stream it
is launched on): the bytes that launch has to read / its duration, against the 8 TB/s HBM peak and against a measured
plain stream-read ceiling; the layout-independent algorithmic figure of SURVEY.md section 8(d) beside it.  cpu_baseline:
oracle/roaring_port.c (the reference's algorithm over roaring-format containers, OpenMP) on a bounded sample of the same
store, rank 0 / N = 1 only.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path[:0] = [ROOT, os.path.join(ROOT, "lapis-silo_amd")]

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak, /opt/skills/guides/MI355X_MICROARCH.md
QUERY_LINEAGE = "B.1"
N_LINEAGES = 2000


def log(*args):
    print(*args, file=sys.stderr, flush=True)


def load_reference_genomes(with_genes=False):
    path = os.path.join(ROOT, "tests", "golden", "exampleDataset", "reference_genomes.json")
    genomes = json.load(open(path))
    # nucleotide segment "main" (the metric is the nucleotide Mutations scan); the 12 genes only for the amino-acid leg
    return {"nucleotideSequences": [g for g in genomes["nucleotideSequences"] if g["name"] == "main"],
            "genes": genomes["genes"] if with_genes else []}


def build_engine(n_sequences, rank, world, all_reduce, device, sharded=False, with_genes=False, with_metadata=False, nuc_positions=None,
                 lineage_order=False, two_pass=False, by_position=True, seed=None, options=None):
    """An engine with one partition of `n_sequences` synthetic rows.  sharded / world > 1: this rank's shard of a sharded
    database — by position range (every rank holds all rows for its window of positions) or by sequence id (by_position
    False: every rank holds all positions of ITS n_sequences rows, generated from `seed`)."""
    from silo_amd import alphabet, synth
    from silo_amd.engine import Engine

    genomes = load_reference_genomes(with_genes)
    if nuc_positions is not None:  # an amino-acid measurement: keep only a stub of the nucleotide store
        genomes["nucleotideSequences"][0]["sequence"] = genomes["nucleotideSequences"][0]["sequence"][:nuc_positions]
    reference = np.array([alphabet.NUCLEOTIDE.char_to_symbol[c] for c in genomes["nucleotideSequences"][0]["sequence"]], dtype=np.uint8)
    seed = synth.DEFAULT_SEED if seed is None else seed
    tree = synth.make_lineage_tree(N_LINEAGES)
    lineage = synth.assign_lineages(n_sequences, tree, seed)
    if lineage_order:
        # rows laid out lineage by lineage, sublineages behind their parent (the reference partitions by Pango lineage and
        # groups its rows by that key, preprocessor.cpp:159-227): a lineage-with-sublineages filter is then a row range
        rank_of = np.empty(len(tree.names), dtype=np.int64)
        rank_of[sorted(range(len(tree.names)), key=lambda k: [int(part) for part in tree.names[k].split(".")[1:]])] = np.arange(len(tree.names))
        lineage = lineage[np.argsort(rank_of[lineage], kind="stable")]
    # the table of lineage substitutions is the database's (every shard has it); the rows are the shard's (`seed`)
    model = synth.make_model(n_sequences, reference, "nuc", tree, lineage, seed=seed, table_seed=synth.DEFAULT_SEED)
    engine = Engine(genomes, device=device)
    if two_pass:  # the generator runs twice per store: counted, then written straight into the adaptive planes (no build-time planes)
        engine.set_option("two_pass_build", 1)
    for name, value in (options or {}).items():  # silo_engine_set_option: how THIS engine's stores are laid out
        engine.set_option(name, value)
    if getattr(build_engine, "comm", None) is not None and (world > 1 or sharded):
        engine.set_comm(build_engine.comm, by_position)  # native RCCL all-reduce / broadcast on the engine's streams
    elif world > 1 or sharded:
        engine.set_sharding(rank, world, by_position, all_reduce)
        if by_position and getattr(build_engine, "broadcast", None) is not None:
            engine.set_broadcast(build_engine.broadcast)
    partition = engine.add_partition(n_sequences)
    window = engine.position_window("main", False)
    engine.generate_synthetic(partition, "main", False, model, window)
    for index, gene in enumerate(genomes["genes"]):
        gene_reference = np.array([alphabet.AMINO_ACID.char_to_symbol[c] for c in gene["sequence"]], dtype=np.uint8)
        gene_model = synth.make_model(n_sequences, gene_reference, "aa", tree, lineage, seed=seed, store_index=index + 1, table_seed=synth.DEFAULT_SEED)
        engine.generate_synthetic(partition, gene["name"], True, gene_model, engine.position_window(gene["name"], True))
    engine.set_lineage_column_ids(partition, "pango_lineage", tree.names, lineage)
    if with_metadata:
        add_synthetic_metadata(engine, partition, n_sequences)
    engine.finalize()
    return engine, model, tree, lineage, window


def make_query():
    return json.dumps({
        "action": {"type": "Mutations", "minProportion": 0.05},
        "filterExpression": {"type": "PangoLineage", "column": "pango_lineage", "value": QUERY_LINEAGE, "includeSublineages": True},
    })


def time_kernel(engine, tree, window, reps):
    """Average duration of one Mutations scan over this rank's window (HIP events on the null stream, where the scan is
    launched), and of every kernel launch inside it (time_kernel.per_launch: plane scans, escape-key pass, runs of the
    missing symbol, sparse keys — each with the bytes it has to read)."""
    import ctypes

    from silo_amd import binding

    store = engine.partition_store(0)
    lib = binding.load_library()
    member = tree.subtree(tree.names.index(QUERY_LINEAGE))
    filt = ctypes.c_void_p()
    binding._check(lib.silo_gpu_bitset_alloc(store.handle, ctypes.byref(filt)))
    binding._check(lib.silo_gpu_bitset_from_lineages(store.handle, filt, member.ctypes.data_as(ctypes.c_void_p), len(member), None))
    n_positions = window[1] - window[0]
    counts = ctypes.c_void_p()
    binding._check(lib.silo_gpu_malloc(4 * n_positions * 5, ctypes.byref(counts)))
    binding._check(lib.silo_gpu_memset_async(counts, 0, 4 * n_positions * 5, None))
    start, stop = binding.GpuEvent(), binding.GpuEvent()
    for _ in range(2):
        binding._check(lib.silo_gpu_mutations_scan(store.handle, 0, filt, 0, n_positions, counts, None))
    start.record()
    for _ in range(reps):
        binding._check(lib.silo_gpu_mutations_scan(store.handle, 0, filt, 0, n_positions, counts, None))
    stop.record()
    ms = start.elapsed_ms(stop) / reps
    kernel = lib.silo_gpu_last_scan_kernel().decode()
    # every launch by itself: HIP events around each launch of `reps` more scans (on the stream it is launched on), averaged
    # per kernel instantiation
    launches = {}
    previous = lib.silo_gpu_tune(7, 1)  # SILO_GPU_TUNE_SCAN_TIMING
    try:
        for _ in range(reps):
            binding._check(lib.silo_gpu_mutations_scan(store.handle, 0, filt, 0, n_positions, counts, None))
            for entry in binding.scan_timings():
                slot = launches.setdefault(entry["kernel"], dict(kernel=entry["kernel"], plane_rows=entry["plane_rows"], bytes=entry["bytes"],
                                                                 blocks=entry["blocks"], ms=[]))
                slot["ms"].append(entry["ms"])
    finally:
        lib.silo_gpu_tune(7, previous)
    per_launch = sorted(({**v, "launches": len(v["ms"]), "ms": sum(v["ms"]) / len(v["ms"])} for v in launches.values()), key=lambda v: -v["ms"])
    time_kernel.per_launch = per_launch
    # the same once more with every pass on the caller's stream (SILO_GPU_TUNE_SIDE_STREAM = 2): the launches one behind the
    # other, each with the device to itself — a launch's own rate, where the figures above are its rate while it shares the
    # HBM with the passes on the side streams
    alone = {}
    previous_side = lib.silo_gpu_tune(5, 2)
    previous = lib.silo_gpu_tune(7, 1)
    try:
        for _ in range(reps):
            binding._check(lib.silo_gpu_mutations_scan(store.handle, 0, filt, 0, n_positions, counts, None))
            for entry in binding.scan_timings():
                alone.setdefault(entry["kernel"], []).append(entry["ms"])
    finally:
        lib.silo_gpu_tune(7, previous)
        lib.silo_gpu_tune(5, previous_side)
    time_kernel.alone_ms = {kernel: sum(v) / len(v) for kernel, v in alone.items()}
    return ms, kernel, store, filt, counts


time_kernel.per_launch = []
time_kernel.alone_ms = {}


def scan_bytes(lib, handle, seqstore_id, positions, w8):
    """Bytes one scan of a sequence store has to read, each once: plane rows + the filter row, 4 per escape key, 12 per run
    of the missing symbol and 8 per sparse key (the last two only where the store derives a symbol)."""
    rows = int(lib.silo_gpu_store_scan_rows(handle, seqstore_id, 0, positions))
    keys = int(lib.silo_gpu_store_scan_escapes(handle, seqstore_id))
    runs = int(lib.silo_gpu_store_scan_runs(handle, seqstore_id))
    sparse = int(lib.silo_gpu_store_scan_sparse_keys(handle, seqstore_id))
    return dict(plane_rows=rows, escape_keys=keys, missing_runs=runs, sparse_keys=sparse, bytes=rows * w8 + w8 + 4 * keys + 12 * runs + 8 * sparse)


LAYOUT_TEXT = {
    0: ("derived symbols: at most positions ONE valid symbol has nearly every row, and that symbol is stored nowhere — its count under a filter is the "
        "filter's cardinality minus the rows of the filter without a valid symbol there (runs of the missing symbol, ambiguity codes) minus the other "
        "symbols' counts, as the reference rebuilds the bitmap it deletes (position.cpp:102-127, mutations.cpp:74-95); one-hot rows only for a second / "
        "third frequent symbol, every other row an escape key"),
    1: ("adaptive planes, chosen per position at finalize: ONE one-hot row of the position's most frequent valid symbol (two or three rows where a "
        "second / third symbol is frequent), or 2 code planes, the other rows as escape keys; positions where neither pays keep 3 identity planes"),
    2: "adaptive code planes: 2 planes per position (codes 1..3 = the 3 most frequent valid symbols) with the other rows as escape keys",
    3: "bit-sliced identity planes: 3 code planes per position instead of 5 one-hot symbol planes",
}


def roofline_of(lib, store, window, sequences, kernel_ms, ceiling_gbps):
    """The roofline block of the line from time_kernel's measurements: the dominant launch (longest by HIP events) by its own
    bytes and duration, every launch beside it, the whole scan, the algorithmic figure of SURVEY.md section 8(d)."""
    w8 = 8 * ((sequences + 63) // 64)
    n_local = window[1] - window[0]
    physical = scan_bytes(lib, store.handle, 0, n_local, w8)
    alg_bytes = n_local * 5 * w8 + w8
    launches = [dict(kernel=v["kernel"], ms=v["ms"], bytes=v["bytes"], GBps=v["bytes"] / (v["ms"] * 1e-3) / 1e9, plane_rows=v["plane_rows"], blocks=v["blocks"],
                     ms_alone=time_kernel.alone_ms.get(v["kernel"]),
                     GBps_alone=v["bytes"] / (time_kernel.alone_ms[v["kernel"]] * 1e-3) / 1e9 if time_kernel.alone_ms.get(v["kernel"]) else None)
                for v in time_kernel.per_launch]
    # the dominant launch = the one that moves the most bytes (a launch on the side stream may LAST longer beside the others without being the work)
    dominant = max(time_kernel.per_launch, key=lambda v: v["bytes"]) if time_kernel.per_launch else dict(
        kernel="?", plane_rows=0, bytes=physical["bytes"], ms=kernel_ms, blocks=0, launches=0)
    achieved = dominant["bytes"] / (dominant["ms"] * 1e-3) / 1e9
    scan_planes = int(lib.silo_gpu_store_scan_planes(store.handle, 0))
    # HBM traffic per launch is a PMC figure (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes, corrected as the
    # microarchitecture guide prescribes); counters cannot be read from inside this process, so the value is the one on file
    # for exactly this launch shape — labelled with its source — or null.
    traffic, traffic_source = None, None
    try:
        pmc = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))
        for key, entry in pmc["kernels"].items():
            name, _, grid = key.rpartition("@")
            if name == dominant["kernel"] and grid == str(dominant["blocks"]) and entry.get("sequences") == sequences and entry.get("bytes") == dominant["bytes"]:
                traffic = entry["hbm_bytes"]
                traffic_source = "profiles/pmc_traffic.json (rocprofv3 --pmc passes of this launch shape in an earlier run; not measured by this run)"
    except (OSError, ValueError, KeyError):
        pass
    return {
        "bound": "hbm",
        "achieved": achieved,
        "peak": HBM_PEAK_GBS,
        "unit": "GB/s",
        "frac": achieved / HBM_PEAK_GBS,
        "traffic": traffic,
        "traffic_source": traffic_source,
        "stream_ceiling_GBps": ceiling_gbps,
        "frac_of_ceiling": achieved / ceiling_gbps if ceiling_gbps else None,
        "kernel": dominant["kernel"],
        "kernel_ms": dominant["ms"],
        "kernel_ms_alone": time_kernel.alone_ms.get(dominant["kernel"]),
        "frac_alone": (dominant["bytes"] / (time_kernel.alone_ms[dominant["kernel"]] * 1e-3) / 1e9 / HBM_PEAK_GBS
                       if time_kernel.alone_ms.get(dominant["kernel"]) else None),
        "kernel_launches_timed": dominant.get("launches", 0),
        "kernel_bytes_per_launch": dominant["bytes"],
        "launches_per_scan": launches,
        "scan_ms": kernel_ms,
        "scan_physical": physical,
        "scan_GBps": physical["bytes"] / (kernel_ms * 1e-3) / 1e9,
        "scan_frac": physical["bytes"] / (kernel_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
        "algorithmic_bytes_per_scan": alg_bytes,
        "algorithmic_GBps": alg_bytes / (kernel_ms * 1e-3) / 1e9,
        "byte_reduction": alg_bytes / physical["bytes"],
        "plane_rows_per_position": physical["plane_rows"] / max(1, n_local),
        "layout": LAYOUT_TEXT.get(scan_planes, LAYOUT_TEXT[3]),
        "note": "frac = the dominant launch's own bytes / its own duration / peak, in the product's configuration: the escape-key pass and the passes for "
                "derived symbols run beside it on side streams and share the HBM (frac_alone / ms_alone: the same launches one behind the other on one "
                "stream); scan_frac = all bytes of the scan / the scan's duration; algorithmic bytes / time exceeds the peak precisely because the "
                "scan moves `byte_reduction` x fewer bytes than the one-hot model of SURVEY.md section 8(d); `also_identity_planes` is the same "
                "query where nothing of that applies (the floor)",
    }


def stream_ceiling(lib, gigabytes=8, reps=5):
    """Measured plain stream-read rate of this device in GB/s (silo_gpu_stream_read_probe: a uint64 sum kernel, 16-byte non-temporal loads)."""
    import ctypes

    from silo_amd import binding

    ms = ctypes.c_float()
    size = int(gigabytes * 1e9)
    binding._check(lib.silo_gpu_stream_read_probe(size, reps, ctypes.byref(ms)))
    return (size & ~15) / (ms.value * 1e-3) / 1e9


def cpu_share():
    """CPU threads this process may really use: affinity mask and cgroup quota, not the host's core count."""
    share = len(os.sched_getaffinity(0))
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            share = min(share, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return max(1, share)


def cpu_baseline(model, tree, lineage, n_sequences, store, filt, budget_positions):
    """The reference's algorithm (oracle/roaring_port.c: roaring-format containers, OpenMP over positions in grains of 300 as
    in mutations.cpp:139-164) on a bounded sample of the same synthetic store, SURVEY.md §8(d) protocol: ONE of the <= 32
    partitions the reference splits a database into (a row range of N/32 sequences: the reference scans its partitions one
    after the other inside every query, query_engine.cpp:40-49), one full 300-position grain per thread, one warm-up pass
    and the best of 5.  Its counts are checked against the GPU's on the same rows and positions.  A second figure times
    the dense CPU oracle (numpy over the character matrix, one thread) for context."""
    import ctypes

    from oracle import cpu_port, dense
    from oracle import synth as oracle_synth
    from silo_amd import binding

    threads = max(1, min(cpu_port.max_threads(), cpu_share()))
    os.environ["OMP_NUM_THREADS"] = str(threads)
    partitions = 32 if n_sequences >= 32 * 65536 else 1
    rows = n_sequences // partitions
    grain = 300
    n_positions = min(model.positions // 64 * 64, budget_positions or threads * grain)
    begin = ((model.positions - n_positions) // 2) // 64 * 64
    t0 = time.time()
    port = cpu_port.PortStore(rows, begin, n_positions, "nuc", model=model)
    build_s = time.time() - t0
    member = tree.subtree(tree.names.index(QUERY_LINEAGE))
    mask = member[lineage[:rows]].astype(bool)
    port_filter = cpu_port.Filter(dense.pack_bits(mask), rows)
    port.mutations_scan(port_filter, n_threads=threads, grain=grain)  # warm-up
    best = None
    counts = None
    for _ in range(5):
        counts, seconds = port.mutations_scan(port_filter, n_threads=threads, grain=grain)
        best = seconds if best is None else min(best, seconds)
    # parity at full size: the GPU's counts for the same rows (the query's filter restricted to the partition) and positions
    lib = binding.load_library()
    row_words = store.row_words
    partition_filter = np.zeros(row_words, dtype=np.uint64)
    packed = dense.pack_bits(mask)
    partition_filter[:len(packed)] = packed
    filter_dev = ctypes.c_void_p()
    binding._check(lib.silo_gpu_bitset_alloc(store.handle, ctypes.byref(filter_dev)))
    binding._check(lib.silo_gpu_bitset_upload(store.handle, filter_dev, partition_filter.ctypes.data_as(ctypes.c_void_p), row_words, None))
    gpu_counts_dev = ctypes.c_void_p()
    binding._check(lib.silo_gpu_malloc(4 * n_positions * 5, ctypes.byref(gpu_counts_dev)))
    binding._check(lib.silo_gpu_memset_async(gpu_counts_dev, 0, 4 * n_positions * 5, None))
    binding._check(lib.silo_gpu_mutations_scan(store.handle, 0, filter_dev, begin, begin + n_positions, gpu_counts_dev, None))
    gpu_counts = np.empty(n_positions * 5, dtype=np.uint32)
    binding._check(lib.silo_gpu_memcpy_d2h(gpu_counts.ctypes.data_as(ctypes.c_void_p), gpu_counts_dev, gpu_counts.nbytes, None))
    lib.silo_gpu_free(gpu_counts_dev)
    lib.silo_gpu_free(filter_dev)
    if not np.array_equal(gpu_counts.reshape(n_positions, 5), counts[:, :5]):
        raise AssertionError("cpu_baseline: GPU counts differ from the CPU port on the sampled rows and positions")
    census = port.census()
    port.close()
    # context: the dense CPU oracle (oracle/dense.py, numpy compare-and-sum over the character matrix) on a slice of the sample
    dense_rows, dense_positions = min(rows, 100_000), min(n_positions, 300)
    symbols = oracle_synth.symbol_matrix(model, np.arange(dense_rows), np.arange(begin, begin + dense_positions))
    t0 = time.perf_counter()
    dense_counts = dense.mutation_counts(symbols, mask[:dense_rows], [0, 1, 2, 3, 4])
    dense_seconds = time.perf_counter() - t0
    if dense_rows == rows and not np.array_equal(dense_counts, counts[:dense_positions, :5]):
        raise AssertionError("cpu_baseline: the dense oracle and the port disagree")
    return {
        "value": rows * n_positions / best,
        "unit": "positions*sequences/s",
        "cores": threads,
        "kind": "port",
        "sample": f"one of {partitions} partitions ({rows} sequences, rows [0,{rows})) x positions [{begin},{begin + n_positions}) of the same "
                  f"{n_sequences}-sequence store, filter cardinality {int(mask.sum())}, {threads} OpenMP threads, grain {grain} positions "
                  f"({-(-n_positions // grain)} grains), 1 warm-up + best of 5; build {build_s:.1f}s; "
                  f"containers array/bitset/run = {census['arrays']}/{census['bitsets']}/{census['runs']}; counts equal to the GPU's on these rows and positions",
        "seconds": best,
        "note": "the port follows the reference's per-position x symbol and_cardinality and its row-wise missing-symbol probes (mutations.cpp:75-82), "
                "which dominate its time; the GPU/CPU ratio is reported, not claimed as kernel quality",
        "dense_oracle": {"value": dense_rows * dense_positions / dense_seconds, "unit": "positions*sequences/s", "cores": 1,
                         "sample": f"{dense_rows} sequences x {dense_positions} positions, numpy (oracle/dense.py), one pass"},
    }


def filter_query(model, tree, variant=0):
    """BASELINE.json configs[2] / SURVEY.md section 8d C3: And(Or(8 eq), N-Of(3 of 8 eq), Not(Or(8 eq)), Maybe(And(8 eq)))
    over 32 distinct (position, symbol) leaves at the positions where most sequences carry a substitution.
    variant k > 0: the same tree over the NEXT 32 positions of that ranking (leaves distinct from every other variant's:
    a batch of variants reads Q x 32 different columns, so its bytes really come from HBM, not from a cache).  Those
    positions are substituted in small, unrelated lineages — eight of their substitutions under And select nothing —, so
    there the leaves under N-Of and Maybe(And) ask for the REFERENCE symbol of their position (most rows have it) and the
    substitutions stay under Or and Not(Or): every variant selects rows, and its count is checked (tests/test_configs_gpu.py)."""
    carried = model.lineage_symbol != 0xFF                      # [P][L]
    weight = carried.astype(np.float64) @ tree.weights          # share of sequences substituted per position
    positions = np.argsort(-weight, kind="stable")[32 * variant:32 * variant + 32]
    assert len(positions) == 32 and carried[positions].any(axis=1).all(), "not enough substituted positions for this variant"
    leaves = []
    for k, p in enumerate(positions):
        lineages = np.nonzero(carried[p])[0]
        symbol = int(model.lineage_symbol[p][lineages[np.argmax(tree.weights[lineages])]])
        if variant > 0 and (k < 8 or 16 <= k < 24):
            symbol = int(model.reference[p])
        leaves.append({"type": "NucleotideEquals", "position": int(p) + 1, "symbol": "-ACGT"[symbol]})
    # leaves are sorted by how many sequences carry them: the most common 8 go under Maybe(And), the rarest 8 under Not(Or)
    return json.dumps({
        "action": {"type": "Aggregated"},
        "filterExpression": {"type": "And", "children": [
            {"type": "Or", "children": leaves[8:16]},
            {"type": "N-Of", "numberOfMatchers": 3, "matchExactly": False, "children": leaves[16:24]},
            {"type": "Not", "child": {"type": "Or", "children": leaves[24:32]}},
            {"type": "Maybe", "child": {"type": "And", "children": leaves[0:8]}},
        ]},
    })


def filter_workload(engine, model, tree, n_sequences, sync, seconds=2.0):
    """Filter -> Aggregated queries per second: sequential latency and 8 client threads on one engine."""
    import threading

    query = filter_query(model, tree)
    count = engine.execute_query(query)[0]["count"]   # also warms the lineage / sparse-plane caches
    wire = query.encode()
    for _ in range(5):
        engine.execute_text(wire)
    sync()
    # native request threads (silo_engine_run_clients), one query at a time each: the interpreter's lock is not in the figure
    expected = engine.execute_text(wire)[1]
    per_second, body = engine.run_clients(wire, 1, seconds)
    if body != expected:
        raise AssertionError("a client's response differs")
    sequential = 1.0 / per_second
    concurrent, body = engine.run_clients(wire, 8, seconds)
    if body != expected:
        raise AssertionError("a client's response differs")
    w8 = 8 * ((n_sequences + 63) // 64)

    # SURVEY.md §8(d) C3 "batched throughput, >= 64 in flight": 64 DIFFERENT queries of this shape (disjoint leaf sets:
    # 2048 columns = 2.56 GB at 10 M sequences) as one silo_engine_execute_batch — parse and compile of all 64 on the
    # host, then ONE k_filter_eval_batch launch for the 64 bit-programs, the counts back in one copy.
    batch_size = 64
    batch = [filter_query(model, tree, variant).encode() for variant in range(batch_size)]
    one_by_one = [engine.execute_text(q) for q in batch]     # also decodes the 2048 one-hot leaf planes into the cache
    if engine.execute_batch_text(batch) != one_by_one:
        raise AssertionError("batched filter queries differ from one-by-one execution")
    sync()
    reps = 0
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < seconds:
        engine.execute_batch_text(batch)
        reps += 1
    batch_seconds = (time.perf_counter() - t0) / reps
    batch_counts = [json.loads(body.decode())["queryResult"][0]["count"] for _, body in one_by_one]

    # the same with 8 request threads, each submitting such batches on its own stream: the host work of one client's batch
    # (parse and compile, spread over the engine's batch workers) overlaps the launches of the others
    n_clients = max(2, min(8, cpu_share()))
    batches_done = []

    def batch_client():
        k = 0
        end = time.perf_counter() + seconds
        while time.perf_counter() < end:
            engine.execute_batch_text(batch)
            k += 1
        batches_done.append(k)

    threads = [threading.Thread(target=batch_client) for _ in range(n_clients)]
    t0 = time.perf_counter()
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    batched_concurrent = sum(batches_done) * batch_size / (time.perf_counter() - t0)
    kernel = filter_batch_kernel_time(engine, model, tree, batch, n_sequences)
    return {
        "workload": f"BASELINE.json configs[2]: And(Or(8), 3-of-8, Not(Or(8)), Maybe(And(8))) over 32 NucleotideEquals leaves "
                    f"-> Aggregated, {n_sequences} sequences",
        "count": count,
        "latency_us": sequential * 1e6,
        "queries_per_s_1_client": 1.0 / sequential,
        "queries_per_s_8_clients": concurrent,
        "algorithmic_bytes_per_query": 32 * w8,
        "batched": {
            "workload": f"{batch_size} different queries of this shape (disjoint leaf sets) in one silo_engine_execute_batch",
            "queries_in_flight": batch_size,
            "ms_per_batch": batch_seconds * 1e3,
            "queries_per_s": batch_size / batch_seconds,
            "us_per_query": batch_seconds / batch_size * 1e6,
            "client_threads": n_clients,
            "queries_per_s_all_clients": batched_concurrent,
            "GBps_all_clients": batched_concurrent * 32 * w8 / 1e9,
            "frac_of_peak_all_clients": batched_concurrent * 32 * w8 / 1e9 / HBM_PEAK_GBS,
            "nonzero_counts": sum(1 for c in batch_counts if c > 0),
            "roofline": kernel,
        },
    }


def filter_batch_kernel_time(engine, model, tree, batch, n_sequences, reps=20):
    """The dominant kernel of the batched filter leg on its own: k_filter_eval_batch over the same 64 programs, lowered by
    hand the way host/operators.cpp lowers the tree (OR_N / CNT_ADD_N / AND_N runs of 8 leaves), timed with HIP events on
    the stream it is launched on.  Bytes: 32 leaf columns per program, nothing written (count only)."""
    import ctypes

    from silo_amd import binding as b

    lib = b.load_library()
    store = engine.partition_store(0)
    programs = []
    planes = []
    symbols = "-ACGTRYSWKMBDHVN"
    for wire in batch:
        children = json.loads(wire.decode())["filterExpression"]["children"]
        groups = [children[0]["children"], children[1]["children"], children[2]["child"]["children"], children[3]["child"]["children"]]
        leaves = []
        for group_index, group in enumerate(groups):
            for leaf in group:
                plane = ctypes.c_void_p()
                b._check(lib.silo_gpu_bitset_alloc(store.handle, ctypes.byref(plane)))
                symbol = symbols.index(leaf["symbol"])
                b._check(lib.silo_gpu_store_sparse_plane(store.handle, 0, leaf["position"] - 1, symbol, plane, None))
                # (the Maybe group reads the symbol's plane alone here: the IUPAC planes it ORs in are built once and
                # cached by the engine as ONE combined column, so the launch reads 32 columns either way)
                leaves.append(plane)
                planes.append(plane)
        code = (b.encode(b.OP_OR_N, 0, imm=0 | (8 << 16))
                + b.encode(b.OP_ZERO, 2) + b.encode(b.OP_ZERO, 3) + b.encode(b.OP_ZERO, 4) + b.encode(b.OP_ZERO, 5)
                + b.encode(b.OP_CNT_ADD_N, 2, 0, 4, imm=8 | (8 << 16)) + b.encode(b.OP_CNT_GE, 1, 2, 4, imm=3) + b.encode(b.OP_AND, 0, 0, 1)
                + b.encode(b.OP_OR_N, 1, imm=16 | (8 << 16)) + b.encode(b.OP_ANDNOT, 0, 0, 1)
                + b.encode(b.OP_AND_N, 1, imm=24 | (8 << 16)) + b.encode(b.OP_AND, 0, 0, 1))
        programs.append((code, leaves, 6))
    prepared = b.PreparedPrograms(programs)  # marshalled once: the timed loop is the C call alone
    binding_counts = prepared.launch(store.handle)
    start, stop = b.GpuEvent(), b.GpuEvent()
    start.record()
    for _ in range(reps):
        prepared.launch(store.handle)
    stop.record()
    ms = start.elapsed_ms(stop) / reps
    for plane in planes:
        lib.silo_gpu_free(plane)
    w8 = 8 * ((n_sequences + 63) // 64)
    bytes_per_launch = len(programs) * 32 * w8
    gbps = bytes_per_launch / (ms * 1e-3) / 1e9
    traffic, traffic_source = None, None
    try:  # PMC figure of this launch shape on file (counters cannot be read in-process): labelled with its source, or null
        entry = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))["kernels"].get("k_filter_eval_batch@64x306")
        if entry is not None and entry.get("programs") == len(programs) and entry.get("sequences") == n_sequences:
            traffic = entry["hbm_bytes"]
            traffic_source = "profiles/pmc_traffic.json (rocprofv3 --pmc FETCH_SIZE of tools/filter_batch_probe.py, same launch shape; not measured by this run)"
    except (OSError, ValueError, KeyError):
        pass
    return {
        "bound": "hbm", "kernel": "k_filter_eval_batch", "programs_per_launch": len(programs),
        "ms_per_launch": ms, "timed": "HIP events around silo_gpu_filter_eval_batch: program-table upload, counter memset, the kernel, count copy and its wait",
        "bytes_per_launch": bytes_per_launch, "achieved": gbps, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gbps / HBM_PEAK_GBS,
        "traffic": traffic, "traffic_source": traffic_source, "counts_nonzero": sum(1 for c in binding_counts if c > 0),
    }


def batch_workload(engine, positions, n_sequences, sync, reps=5):
    """8 concurrent Mutations queries with different lineage filters: one by one vs. one silo_engine_execute_batch
    call in which their scans share a single pass over the planes (K1c)."""
    queries = [json.dumps({
        "action": {"type": "Mutations", "minProportion": 0.05},
        "filterExpression": {"type": "PangoLineage", "column": "pango_lineage", "value": lineage, "includeSublineages": True},
    }).encode() for lineage in ("B.1", "B.2", "B.3", "B.1.1", "B.1.2", "B.1.3", "B.2.1", "B.2.2")]
    one_by_one = [engine.execute_text(q) for q in queries]
    batched = engine.execute_batch_text(queries)
    if batched != one_by_one:
        raise AssertionError("batched results differ from one-by-one results")
    sync()
    t0 = time.perf_counter()
    for _ in range(reps):
        for q in queries:
            engine.execute_text(q)
    sync()
    sequential = (time.perf_counter() - t0) / reps
    t0 = time.perf_counter()
    for _ in range(reps):
        engine.execute_batch_text(queries)
    sync()
    together = (time.perf_counter() - t0) / reps
    return {
        "workload": f"{len(queries)} Mutations queries (PangoLineage B.1*, B.2*, B.3*, B.1.1*, B.1.2*, B.1.3*, B.2.1*, B.2.2*), {n_sequences} sequences; "
                    "responses identical both ways",
        "ms_one_by_one": sequential * 1e3,
        "ms_one_batch": together * 1e3,
        "value_one_batch": len(queries) * n_sequences * positions / together,
        "unit": "positions*sequences/s",
    }


def config4_queries(tree, reference_text):
    """BASELINE.json configs[4]: 100 distinct lineage filters (every fourth AND a nucleotide predicate), each with a Mutations
    and an AminoAcidMutations action — 200 queries."""
    queries = []
    for k, name in enumerate(tree.names[1:101]):
        expression = {"type": "PangoLineage", "column": "pango_lineage", "value": name, "includeSublineages": True}
        if k % 4 == 3:
            position = 1000 + 257 * k
            expression = {"type": "And", "children": [expression, {"type": "NucleotideEquals", "position": position, "symbol": reference_text[position - 1]}]}
        for action in ("Mutations", "AminoAcidMutations"):
            queries.append(json.dumps({"action": {"type": action, "minProportion": 0.05}, "filterExpression": expression}).encode())
    return queries


def config4_workload(engine, tree, reference_text, positions, aa_positions, total_sequences, world, sync, reps=3, reduce_max=None):
    """BASELINE.json configs[4]: `total_sequences` sequences (genome + 12 genes) sharded by sequence id over `world` GPUs (this
    engine holds this rank's rows), ONE silo_engine_execute_batch of the 200 queries: their scans share passes over the
    planes, 8 filters at a time (K1c); with more than one rank every count table is all-reduced (one collective per
    query, enqueued on the engine's stream)."""
    queries = config4_queries(tree, reference_text)
    batched = engine.execute_batch_text(queries)
    if batched[:6] != [engine.execute_text(q) for q in queries[:6]] or any(status != 200 for status, _ in batched):
        raise AssertionError("batched results differ from one-by-one results")
    sync()
    t0 = time.perf_counter()
    for _ in range(reps):
        engine.execute_batch_text(queries)
    sync()
    seconds = (time.perf_counter() - t0) / reps
    if reduce_max is not None:
        seconds = reduce_max(seconds)
    rows = sum(len(json.loads(body.decode())["queryResult"]) for _, body in batched)
    return {
        "workload": f"BASELINE.json configs[4]: {total_sequences} sequences (genome + 12 genes) "
                    + (f"sharded by sequence id over {world} GPUs, " if world > 1 else "on the one GPU, ")
                    + "ONE batch of 100 lineage filters (every fourth AND a nucleotide predicate) x (Mutations + AminoAcidMutations) = 200 queries"
                    + ("; one all-reduce per count table" if world > 1 else ""),
        "queries": len(queries),
        "ms_per_batch": seconds * 1e3,
        "value": (len(queries) // 2) * total_sequences * (positions + aa_positions) / seconds,
        "unit": "positions*sequences/s",
        "scaling": "strong",
        "queries_per_s": len(queries) / seconds,
        "mutation_rows": rows,
    }


def selective_workload(engine, lib, tree, lineage, positions, n_sequences, sync, reps=20):
    """Mutations under SELECTIVE filters (one exact lineage: a few thousand rows scattered over the store): the dense
    scan would cost the same 16 ms whatever the filter selects; K1s gathers only the 64-byte sectors of the planes
    that hold selected rows.  Timed with the routing on (default) and off; responses must be identical."""
    sizes = np.bincount(lineage, minlength=len(tree.names))
    order = [int(i) for i in np.argsort(sizes) if sizes[i] > 0]
    picks = {"smallest lineage": order[0], "median lineage": order[len(order) // 2]}
    out = {}
    for label, index in picks.items():
        query = json.dumps({
            "action": {"type": "Mutations", "minProportion": 0.05},
            "filterExpression": {"type": "PangoLineage", "column": "pango_lineage", "value": tree.names[index], "includeSublineages": False},
        }).encode()
        timings = {}
        responses = {}
        for mode, divisor in (("gather", 0), ("dense", -1)):
            lib.silo_gpu_tune(3, divisor)
            responses[mode] = engine.execute_text(query)
            sync()
            t0 = time.perf_counter()
            for _ in range(reps):
                engine.execute_text(query)
            sync()
            timings[mode] = (time.perf_counter() - t0) / reps * 1e3
        lib.silo_gpu_tune(3, 0)
        if responses["gather"] != responses["dense"] or responses["gather"][0] != 200:
            raise AssertionError("sparse-filter routing changed a response")
        out[label] = {"lineage": tree.names[index], "rows_selected": int(sizes[index]), "ms_per_query": timings["gather"],
                      "ms_per_query_dense_scan": timings["dense"], "mutation_rows": len(json.loads(responses["gather"][1].decode())["queryResult"])}
    return {"workload": f"Mutations under one exact PangoLineage (no sublineages), {n_sequences} sequences", "queries": out}


def metadata_workload(engine, n_sequences, sync, seconds=1.0):
    """SURVEY.md §8(f) row 3 on the 1 M-sequence engine: metadata predicates feeding the filter program (K5) and
    Aggregated with groupByFields (K6); latency per query through executeQuery."""
    lineage_filter = json.loads(make_query())["filterExpression"]
    queries = {
        "filter: country = C7 AND age in [20,40] AND lineage B.1* -> Aggregated": {
            "action": {"type": "Aggregated"},
            "filterExpression": {"type": "And", "children": [
                {"type": "StringEquals", "column": "country", "value": "C7"},
                {"type": "IntBetween", "column": "age", "from": 20, "to": 40}, lineage_filter]}},
        "group by country (50 groups), lineage B.1* filter": {
            "action": {"type": "Aggregated", "groupByFields": ["country"], "orderByFields": ["country"]}, "filterExpression": lineage_filter},
        "group by country x age (5000 groups), all rows": {
            "action": {"type": "Aggregated", "groupByFields": ["country", "age"], "orderByFields": ["country", "age"]},
            "filterExpression": {"type": "True"}},
    }
    out = {}
    for label, query in queries.items():
        wire = json.dumps(query).encode()
        status, body = engine.execute_text(wire)
        if status != 200:
            raise RuntimeError(body.decode())
        rows = json.loads(body.decode())["queryResult"]
        for _ in range(3):
            engine.execute_text(wire)
        sync()
        n = 0
        t0 = time.perf_counter()
        while time.perf_counter() - t0 < seconds:
            engine.execute_text(wire)
            n += 1
        out[label] = {"latency_us": (time.perf_counter() - t0) / n * 1e6, "rows": len(rows),
                      "count": sum(row["count"] for row in rows)}
    return {"workload": f"metadata predicates and group-by, {n_sequences} sequences (synthetic country / age columns)", "queries": out}


def add_synthetic_metadata(engine, partition, n_sequences):
    rng = np.random.default_rng(12345)
    country = rng.integers(0, 50, size=n_sequences)
    age = rng.integers(0, 100, size=n_sequences)
    engine.append_metadata(partition, "country", "indexed_string", [f"C{c}" for c in country])
    engine.append_metadata(partition, "age", "int", [str(a) for a in age])


def run_steps(engine, query, steps, warmup, sync):
    query = query.encode()
    for _ in range(warmup):
        engine.execute_text(query)
    sync()
    t0 = time.perf_counter()
    body = None
    for _ in range(steps):
        status, body = engine.execute_text(query)   # the response body, as silo_api would send it
    sync()
    elapsed = time.perf_counter() - t0
    if status != 200:
        raise RuntimeError(body.decode())
    return elapsed, json.loads(body.decode())["queryResult"]


def launch_ranks(n_gpus):
    """Runs this script under `python -m torch.distributed.run --nproc-per-node N` as a child process."""
    import socket
    import subprocess

    with socket.socket() as probe:
        probe.bind(("127.0.0.1", 0))
        port = probe.getsockname()[1]
    command = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n_gpus), "--master-addr", "127.0.0.1",
               "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    child = subprocess.run(command, stdout=subprocess.PIPE, env=env, text=True)  # stderr passes through
    lines = [line for line in child.stdout.splitlines() if line.startswith("{")]
    if lines:
        print(lines[-1], flush=True)
    return child.returncode if child.returncode != 0 or lines else 1


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--sequences", type=int, default=10_000_000)
    ap.add_argument("--shard", choices=["both", "position", "sequence"], default="both",
                    help="N > 1: the headline database sharded by position range (the north star's split; always measured, it is `value`), "
                         "BASELINE.json configs[4] sharded by sequence id, or both (default)")
    ap.add_argument("--config4-sequences", type=int, default=50_000_000, help="sequences of the configs[4] database (all ranks together)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-also", action="store_true", help="the headline measurement only: no further legs")
    ap.add_argument("--no-client-threads", action="store_true",
                    help="skip the config-2 filter-query legs (8 / 16 client threads; > 100 000 launches make a kernel trace large)")
    ap.add_argument("--rehearse-on-one-gpu", action="store_true",
                    help="testing only: every rank uses GPU 0 and the collectives go through gloo — the multi-rank code path of this file "
                         "on a one-GPU box (RCCL refuses two ranks on one device); the numbers mean nothing")
    ap.add_argument("--force-dist", action="store_true", help="use the torch.distributed / RCCL path even with one rank (testing)")
    ap.add_argument("--cpu-positions", type=int, default=0, help="positions in the CPU baseline sample (0 = sized to ~10-30 s of CPU work)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N` on its own: start the N ranks as CHILD processes (one per GPU, the same launcher the
        # driver uses) before anything in this process has touched the GPU, relay rank 0's JSON line and leave with the
        # launcher's exit code.  Never exec: a process that has initialised the GPU must not replace itself.
        sys.exit(launch_ranks(args.gpus))

    # The contract is ONE JSON line on stdout.  Native libraries write there too (RCCL prints a five-line version banner
    # when the communicator comes up), so file descriptor 1 is pointed at stderr for the whole run and the result line
    # goes to a private copy of the real stdout.
    sys.stdout.flush()
    result_stream = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)

    rank = int(os.environ.get("RANK", "0"))
    local_rank = 0 if args.rehearse_on_one_gpu else int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}")

    all_reduce = None
    comm = None
    dist = None
    torch = None
    use_dist = world > 1 or args.force_dist
    if use_dist:
        # torch FIRST: it bundles its own HIP runtime (and RCCL) under the same SONAMEs as /opt/rocm's; whichever is
        # loaded first serves the whole process, and torch only works with its own.
        import torch
        import torch.distributed as dist

    from silo_amd import binding

    lib = binding.load_library()
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        torch.cuda.set_device(local_rank)
        # control plane (rendezvous, barrier, max over ranks, the communicator id): gloo over TCP.  Data path: the
        # engine's own RCCL communicator (silo_gpu_comm_create -> ncclCommInitRank), whose all-reduce of the count table
        # is enqueued on the engine's HIP stream between the scan kernels and the row selection.
        dist.init_process_group("gloo")
        if args.rehearse_on_one_gpu:
            # testing only: RCCL refuses two ranks on one device, so the collectives go through gloo and host memory,
            # enqueued on (and synchronising) the engine stream they are handed
            import ctypes

            def all_reduce(device_ptr, n, stream):
                host = np.empty(n, dtype=np.int32)  # a wrapping int32 sum has the same bits as a uint32 sum
                binding._check(lib.silo_gpu_memcpy_d2h(host.ctypes.data_as(ctypes.c_void_p), device_ptr, host.nbytes, stream))
                tensor = torch.from_numpy(host)
                dist.all_reduce(tensor)
                binding._check(lib.silo_gpu_memcpy_h2d(device_ptr, host.ctypes.data_as(ctypes.c_void_p), host.nbytes, stream))

            def broadcast(device_ptr, nbytes, root, stream):
                host = np.empty(nbytes, dtype=np.uint8)
                if rank == root:
                    binding._check(lib.silo_gpu_memcpy_d2h(host.ctypes.data_as(ctypes.c_void_p), device_ptr, nbytes, stream))
                tensor = torch.from_numpy(host)
                dist.broadcast(tensor, src=root)
                if rank != root:
                    binding._check(lib.silo_gpu_memcpy_h2d(device_ptr, host.ctypes.data_as(ctypes.c_void_p), nbytes, stream))

            build_engine.broadcast = broadcast
        else:
            unique_id = [binding.comm_unique_id() if rank == 0 else None]
            dist.broadcast_object_list(unique_id, src=0)
            comm = binding.Comm(unique_id[0], rank, world, local_rank)  # ncclCommInitRank: collective over the ranks
            build_engine.comm = comm

        def sync():
            dist.barrier()
            torch.cuda.synchronize()

        def reduce_max(seconds):
            t = torch.tensor([seconds], dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            return float(t.item())
    else:
        def sync():
            binding._check(lib.silo_gpu_stream_synchronize(None))

        def reduce_max(seconds):
            return seconds

    collective = ("through gloo and host memory (one-GPU rehearsal)" if comm is None else "by silo_gpu_allreduce_counts (native ncclAllReduce on the engine stream)")
    t0 = time.time()
    # ONE database with the nucleotide genome AND the 12 genes resident together (BASELINE.json configs[3]); N > 1: sharded
    # by position range — every rank holds all rows for its window of the genome and of every gene
    engine, model, tree, lineage, window = build_engine(args.sequences, rank, world, all_reduce, local_rank, use_dist, with_genes=True)
    log(f"[rank {rank}] store ready in {time.time() - t0:.1f}s: {args.sequences} sequences, positions {window}, "
        f"{engine.partition_store(0).device_bytes / 1e9:.1f} GB in HBM")
    query = make_query()
    positions = model.positions
    genes = load_reference_genomes(True)["genes"]
    aa_positions = sum(len(g["sequence"]) for g in genes)
    aa_query = json.dumps({"action": {"type": "AminoAcidMutations", "minProportion": 0.05},
                           "filterExpression": json.loads(query)["filterExpression"]})

    elapsed, rows = run_steps(engine, query, args.steps, args.warmup, sync)
    elapsed = reduce_max(elapsed)
    filter_us, action_us = engine.last_timings()
    ms_per_step = elapsed / args.steps * 1e3
    value = args.sequences * positions / (elapsed / args.steps)

    # roofline of the dominant kernel, on this rank's window
    w8 = 8 * ((args.sequences + 63) // 64)
    kernel_ms, kernel_name, store, filt, counts_dev = time_kernel(engine, tree, window, reps=max(5, args.steps))
    ceiling = stream_ceiling(lib) if rank == 0 else None
    result = {
        "metric": "Mutations-scan positions*sequences/s",
        "value": value,
        "unit": "positions*sequences/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": ms_per_step,
        "higher_is_better": True,
        "scaling": "strong",
        "vs_baseline": None,
        "dtype": "u64",
        "data": "synthetic",
        "config": {
            "workload": f"{args.sequences}-sequence x {positions}-nt nucleotide Mutations scan (minProportion 0.05) with a "
                        f"PangoLineage sublineage filter ({QUERY_LINEAGE}*), through QueryEngine::executeQuery",
            "sequences": args.sequences,
            "positions": positions,
            "mutation_rows": len(rows),
            "sharding": "none" if not use_dist else f"position-range x{world}, all-reduce of counts[{positions}][5] " + collective,
            "reference_phases_us": {"filter": filter_us, "action": action_us},
        },
        "roofline": roofline_of(lib, store, window, args.sequences, kernel_ms, ceiling),
    }

    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        # ~10-30 s of CPU work in all (build + 6 passes): one partition, one 300-position grain per thread
        try:
            result["cpu_baseline"] = cpu_baseline(model, tree, lineage, args.sequences, store, filt, args.cpu_positions)
        except AssertionError:
            raise
        except Exception as error:  # the baseline is informational; a failure to build it must not hide the GPU number
            result["cpu_baseline"] = {"value": None, "unit": "positions*sequences/s", "cores": 0, "kind": "port", "sample": f"failed: {error}"}

    def aa_physical(engine, window_of=None):
        handle = engine.partition_store(0).handle
        total = dict(plane_rows=0, escape_keys=0, missing_runs=0, sparse_keys=0, bytes=0)
        per_gene = {}
        for g in genes:
            begin, end = engine.position_window(g["name"], True)
            part = scan_bytes(lib, handle, engine.seqstore_id(0, g["name"], True), end - begin, w8)
            per_gene[g["name"]] = part["plane_rows"]
            for key in total:
                total[key] += part[key]
        return total, per_gene

    if not args.no_also:
        # BASELINE.json configs[3], amino-acid leg, on the SAME database (N > 1: on its position shards, one all-reduce per query)
        elapsed_aa, rows_aa = run_steps(engine, aa_query, args.steps, args.warmup, sync)
        elapsed_aa = reduce_max(elapsed_aa)
        physical_aa, rows_per_gene = aa_physical(engine)
        result["also_amino_acid_full"] = {
            "workload": f"AminoAcidMutations over all 12 genes (BASELINE.json configs[3], amino-acid leg) on the same {args.sequences}-sequence database, same filter"
                        + (f"; position-range x{world}" if use_dist else ""),
            "value": args.sequences * aa_positions / (elapsed_aa / args.steps),
            "unit": "positions*sequences/s",
            "ms_per_step": elapsed_aa / args.steps * 1e3,
            "algorithmic_GBps_whole_query": aa_positions * 22 * w8 / (elapsed_aa / args.steps) / 1e9,
            "plane_rows_per_gene": rows_per_gene,
            "physical_this_rank": physical_aa,
            "physical_GBps_whole_query_this_rank": physical_aa["bytes"] / (elapsed_aa / args.steps) / 1e9,
            "mutation_rows": len(rows_aa),
        }
        result["config"]["database"] = (f"one database: nucleotide genome + 12 genes + the runs of the missing symbol, "
                                        f"{engine.partition_store(0).device_bytes / 1e9:.1f} GB of HBM" + (" on this rank" if use_dist else " on one GPU"))
    if rank == 0 and world == 1 and not use_dist and not args.no_also:
        if not args.no_client_threads:
            result["filter_queries"] = filter_workload(engine, model, tree, args.sequences, sync)
        result["batched_queries"] = batch_workload(engine, positions, args.sequences, sync)
        result["selective_queries"] = selective_workload(engine, lib, tree, lineage, positions, args.sequences, sync)
    lib.silo_gpu_free(filt)
    lib.silo_gpu_free(counts_dev)
    engine.close()

    reference_text = load_reference_genomes(False)["nucleotideSequences"][0]["sequence"]
    if not args.no_also and (world == 1 or args.shard in ("both", "sequence")):
        # BASELINE.json configs[4]: the big database, sharded by sequence id (each rank generates ITS rows), built in two passes
        per_rank = args.config4_sequences // world
        t_build = time.perf_counter()
        engine4, model4, tree4, _, _ = build_engine(per_rank, rank, world, all_reduce, local_rank, use_dist, with_genes=True, two_pass=per_rank > 12_000_000,
                                                    by_position=False, seed=0x5110C0DE + 7919 * rank)
        build_seconds = time.perf_counter() - t_build
        leg = config4_workload(engine4, tree4, reference_text, positions, aa_positions, per_rank * world, world, sync, reps=3, reduce_max=reduce_max)
        leg["device_GB_this_rank"] = engine4.partition_store(0).device_bytes / 1e9
        leg["build_seconds"] = build_seconds
        if use_dist:
            leg["sharding"] = f"sequence-id x{world} ({per_rank} rows per rank), all-reduce per count table " + collective
        result["also_config4"] = leg
        if use_dist and not args.no_client_threads:
            # filter -> Aggregated on the sequence-id shards: every rank counts its rows, one all-reduce of the count per query
            wire = filter_query(model4, tree4).encode()
            status, body = engine4.execute_text(wire)
            if status != 200:
                raise RuntimeError(body.decode())
            count = json.loads(body.decode())["queryResult"][0]["count"]
            for _ in range(5):
                engine4.execute_text(wire)
            sync()
            n_queries = 300
            t0 = time.perf_counter()
            for _ in range(n_queries):
                engine4.execute_text(wire)
            sync()
            seconds = reduce_max(time.perf_counter() - t0)
            result["filter_queries"] = {
                "workload": f"BASELINE.json configs[2] tree -> Aggregated on the {per_rank * world}-sequence database sharded by sequence id x{world}: "
                            "every rank evaluates the filter on its rows, the counts are all-reduced",
                "count": count,
                "latency_us": seconds / n_queries * 1e6,
                "queries_per_s_1_client": n_queries / seconds,
            }
        engine4.close()

    if rank == 0 and world == 1 and not use_dist and not args.no_also and args.sequences != 1_000_000:
        # BASELINE.json configs[1]: 1 M sequences, same query
        engine1, model1, tree1, lineage1, window1 = build_engine(1_000_000, 0, 1, None, local_rank, with_genes=True, with_metadata=True)  # no collective
        elapsed1, rows1 = run_steps(engine1, query, args.steps, args.warmup, sync)
        elapsed_aa1, rows_aa1 = run_steps(engine1, aa_query, args.steps, args.warmup, sync)
        result["also_amino_acid"] = {
            "workload": "AminoAcidMutations over all 12 genes (BASELINE.json configs[3], amino-acid leg), 1000000 sequences, same filter",
            "value": 1_000_000 * aa_positions / (elapsed_aa1 / args.steps),
            "unit": "positions*sequences/s",
            "ms_per_step": elapsed_aa1 / args.steps * 1e3,
            "mutation_rows": len(rows_aa1),
        }
        kernel_ms1, _, store1, filt1, counts1 = time_kernel(engine1, tree1, window1, reps=max(5, args.steps))
        roofline1 = roofline_of(lib, store1, window1, 1_000_000, kernel_ms1, ceiling)
        result["also"] = {
            "workload": "BASELINE.json configs[1]: 1000000 sequences, same query",
            "value": 1_000_000 * positions / (elapsed1 / args.steps),
            "ms_per_step": elapsed1 / args.steps * 1e3,
            "scan_ms": kernel_ms1,
            "scan_frac": roofline1["scan_frac"],
            "dominant_kernel": roofline1["kernel"],
            "dominant_kernel_frac": roofline1["frac"],
            "mutation_rows": len(rows1),
        }
        result["also_metadata"] = metadata_workload(engine1, 1_000_000, sync)
        lib.silo_gpu_free(filt1)
        lib.silo_gpu_free(counts1)
        engine1.close()
        if args.sequences == 10_000_000:
            # What the headline degrades to when the data is not alignment-like: the same store kept on its 3 identity planes per
            # position (nothing derived, no one-hot rows: the floor), and the layout of round 2 (a one-hot row for the most
            # numerous symbol too) — the same query, the same rows.
            for label, knob, text in (("also_identity_planes", -1, "3 identity code planes per position, every cell read (the floor: what the query costs when no "
                                                                    "position has a dominant symbol)"),
                                      ("also_one_hot_rows", 3, "a one-hot row for the most numerous symbol of every position too (nothing derived: the layout of round 2)")):
                engine_k, _, tree_k, _, window_k = build_engine(args.sequences, 0, 1, None, local_rank, options={"store_layout": knob})
                elapsed_k, rows_k = run_steps(engine_k, query, max(5, args.steps // 2), args.warmup, sync)
                if rows_k != rows:
                    raise AssertionError(f"{label}: the rows differ from the headline's")
                kernel_ms_k, _, store_k, filt_k, counts_k = time_kernel(engine_k, tree_k, window_k, reps=5)
                roofline_k = roofline_of(lib, store_k, window_k, args.sequences, kernel_ms_k, ceiling)
                result[label] = {
                    "workload": f"the headline query on the {args.sequences}-sequence genome stored as: {text}",
                    "value": args.sequences * positions / (elapsed_k / max(5, args.steps // 2)),
                    "unit": "positions*sequences/s",
                    "ms_per_step": elapsed_k / max(5, args.steps // 2) * 1e3,
                    "device_GB": store_k.device_bytes / 1e9,
                    "scan_ms": kernel_ms_k,
                    "scan_physical_bytes": roofline_k["scan_physical"]["bytes"],
                    "scan_frac": roofline_k["scan_frac"],
                    "dominant_kernel": roofline_k["kernel"],
                    "dominant_kernel_ms": roofline_k["kernel_ms"],
                    "dominant_kernel_frac": roofline_k["frac"],
                    "dominant_kernel_frac_of_ceiling": roofline_k["frac_of_ceiling"],
                    "mutation_rows": len(rows_k),
                }
                lib.silo_gpu_free(filt_k)
                lib.silo_gpu_free(counts_k)
                engine_k.close()
            # the same query where the rows lie in lineage order: column tiles and key slices without a selected row are not read
            engine_sorted, _, _, _, _ = build_engine(args.sequences, 0, 1, None, local_rank, lineage_order=True)
            elapsed_sorted, rows_sorted = run_steps(engine_sorted, query, args.steps, args.warmup, sync)
            result["also_rows_in_lineage_order"] = {
                "workload": f"the headline query on a {args.sequences}-sequence genome whose rows lie lineage by lineage (sublineages behind their parent), as the "
                            "reference's partitioning by Pango lineage lays them out; the headline database assigns lineages to rows at random",
                "value": args.sequences * positions / (elapsed_sorted / args.steps),
                "unit": "positions*sequences/s",
                "ms_per_step": elapsed_sorted / args.steps * 1e3,
                "mutation_rows": len(rows_sorted),
            }
            engine_sorted.close()

    if use_dist:
        dist.barrier()
        if comm is not None:
            build_engine.comm = None
            comm.close()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(result), file=result_stream, flush=True)


if __name__ == "__main__":
    main()
