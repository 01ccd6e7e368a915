"""N > 1 path: world_size-2 gloo groups.  The CPU test covers the host sharding logic; the GPU test runs
two real engine ranks on one GPU with a gloo-backed all-reduce callback and compares with one rank."""
import json
import os
import socket
import subprocess
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def launch(world, mode, shard, timeout=600, rows=None):
    port = free_port()
    procs = []
    for rank in range(world):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   OMP_NUM_THREADS="2", HSA_ENABLE_IPC_MODE_LEGACY="0")
        if rows is not None:
            env["SILO_TEST_ROWS"] = str(rows)
        procs.append(subprocess.Popen([sys.executable, os.path.join(HERE, "multi_rank_worker.py"), mode, shard],
                                      stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env))
    outputs = []
    for proc in procs:
        try:
            out, err = proc.communicate(timeout=timeout)
        except subprocess.TimeoutExpired:
            for p in procs:
                p.kill()
            raise
        assert proc.returncode == 0, err[-2000:]
        outputs.append(out)
    return json.loads(outputs[0].strip().splitlines()[-1])


def test_position_windows_and_gloo_all_reduce_cpu(built):
    out = launch(2, "cpu", "position")
    for name, length in (("main", 997), ("S", 211)):
        windows = out[name]["windows"]
        assert windows[0][0] == 0 and windows[-1][1] == length
        assert all(a[1] == b[0] for a, b in zip(windows[:-1], windows[1:]))  # disjoint cover
        assert out[name]["equal"]


@pytest.mark.gpu
@pytest.mark.parametrize("shard", ["position", "sequence"])
def test_two_ranks_match_one_rank_gpu(built, shard):
    single = launch(1, "gpu", shard)
    double = launch(2, "gpu", shard)
    assert [s for s, _ in single["results"]] == [200] * (10 if shard == "position" else 8)
    assert single["results"] == double["results"]
    # and not only with each other: rank 0 ran every query through the CPU oracle on the unsharded data
    assert all(single["matches_oracle"]) and all(double["matches_oracle"]), (single["matches_oracle"], double["matches_oracle"])
    # two ranks that run DIFFERENT queries are refused (500) on both ranks — the all-reduce carries the query's fingerprint —
    # and answer the next common query as before; a single rank has nobody to disagree with
    n_cases = 2 if shard == "position" else 3
    assert double["mismatch_refused"] == [[True] * n_cases + [True]] * 2, double["mismatch_refused"]
    assert single["mismatch_refused"] == [[False] * n_cases + [True]], single["mismatch_refused"]
    if shard == "sequence":  # Details / FastaAligned: the shards' rows, concatenated, are the unsharded rows
        assert single["row_actions_concatenate"] == [True, True] and double["row_actions_concatenate"] == [True, True]


@pytest.mark.gpu
@pytest.mark.parametrize("shard", ["position", "sequence"])
def test_two_ranks_with_re_encoded_stores_match_the_dense_oracle_gpu(built, shard):
    """The same at 140 000 rows (>= 65 536 on every shard): finalize re-encodes the stores — derived symbols, one-hot rows,
    slice-major escape keys, runs of the missing symbol — and the tiled kernels, the escape pass and the derived-symbol passes
    run on every rank: sharding + collective + adaptive layout together.  The restatement of the reference (silo_oracle.py)
    takes ten minutes per query at this size (its row-wise probes of the missing symbol, mutations.cpp:75-82, in Python), so
    the checker is the naive counter (oracle/dense.py over the generator's twin): count tables, thresholds, rows, proportions —
    one by one and as ONE batch of 16 queries (8 filters per pass over rows and keys)."""
    out = launch(2, "gpu_big", shard, timeout=900)
    assert out["rows_per_rank"] == (140_000 if shard == "position" else 70_000) and out["re_encoded"], out["layout"]
    assert all(out["equal"]) and len(out["equal"]) >= 20, out["equal"]
    assert all(out["batch_equal"]) and len(out["batch_equal"]) == 16
    assert out["nonempty"] >= 15


@pytest.mark.gpu
def test_hundred_query_batch_on_sequence_shards_matches_oracle_gpu(built):
    """BASELINE.json configs[4] in shape: ONE silo_engine_execute_batch of 100 queries (lineage filter, every other one
    ANDed with a nucleotide predicate under Not / Maybe, Mutations and AminoAcidMutations) on two sequence-id sharded
    ranks — filters evaluated per shard, the scans of the batch sharing plane passes, one all-reduce per count table —
    against the oracle on the unsharded data."""
    out = launch(2, "batch100", "sequence")
    assert out["queries"] == 100 and all(out["equal"]), [k for k, ok in enumerate(out["equal"]) if not ok]
    assert sum(1 for rows in out["rows"] if rows > 0) >= 80  # most of the hundred answers carry mutation rows


@pytest.mark.gpu
def test_bench_rccl_path_on_one_gpu(built):
    """bench.py --force-dist: the multi-rank code path with ONE rank — gloo for the control plane, the engine's own
    RCCL communicator (silo_gpu_comm_create -> ncclCommInitRank) for the data path: the count table of every query is
    all-reduced by silo_gpu_allreduce_counts (ncclAllReduce, ncclUint32) on the engine's stream, with torch's HIP
    runtime and RCCL serving the process.  Must give the same rows as the plain single-process run."""
    root = os.path.dirname(HERE)
    common = [sys.executable, os.path.join(root, "bench.py"), "--gpus", "1", "--sequences", "200000", "--steps", "2", "--warmup", "1",
              "--no-also", "--no-cpu-baseline"]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(free_port()), HSA_ENABLE_IPC_MODE_LEGACY="0")
    plain = subprocess.run(common, capture_output=True, text=True, env=env, timeout=900)
    assert plain.returncode == 0, plain.stderr[-2000:]
    forced = subprocess.run(common + ["--force-dist"], capture_output=True, text=True, env=env, timeout=900)
    assert forced.returncode == 0, forced.stderr[-2000:]
    # exactly ONE line on stdout, the JSON line — RCCL's version banner and the like go to stderr
    assert len(plain.stdout.strip().splitlines()) == 1 and len(forced.stdout.strip().splitlines()) == 1, forced.stdout[:500]
    a = json.loads(plain.stdout.strip().splitlines()[-1])
    b = json.loads(forced.stdout.strip().splitlines()[-1])
    assert a["config"]["mutation_rows"] == b["config"]["mutation_rows"] > 0
    assert b["config"]["sharding"].startswith("position-range x1") and "silo_gpu_allreduce_counts" in b["config"]["sharding"]
    assert 0 < b["roofline"]["frac"] <= 1.0  # the dominant launch's bytes / its time / peak: a fraction (small at 200 000 rows)
    assert b["roofline"]["kernel"] in {entry["kernel"] for entry in b["roofline"]["launches_per_scan"]} and b["roofline"]["stream_ceiling_GBps"] > 1000


@pytest.mark.gpu
def test_position_window_shards_partition_the_unsharded_result(built):
    """Each rank of a position-range sharded database generates and scans only its slice of the genome
    (bench.py --gpus N).  Without a collective, a rank's Mutations rows are exactly the unsharded rows whose
    position lies in its window; the windows tile the genome."""
    import re

    root = os.path.dirname(HERE)
    sys.path[:0] = [root, os.path.join(root, "lapis-silo_amd")]
    import bench

    n = 150_000
    query = bench.make_query()
    engine, model, tree, lineage, window = bench.build_engine(n, 0, 1, None, 0)
    assert window == (0, model.positions)
    full = engine.execute_query(query)
    count = engine.execute_query({"action": {"type": "Aggregated"}, "filterExpression": json.loads(query)["filterExpression"]})
    engine.close()
    assert len(full) > 100

    def position(row):
        return int(re.match(r"^[^0-9]*([0-9]+)", row["mutation"]).group(1)) - 1

    world = 3
    seen = []
    previous_end = 0
    for rank in range(world):
        shard, _, _, _, shard_window = bench.build_engine(n, rank, world, None, 0, sharded=True)
        assert shard_window[0] == previous_end
        previous_end = shard_window[1]
        rows = shard.execute_query(query)
        assert rows == [row for row in full if shard_window[0] <= position(row) < shard_window[1]]
        # single-filter cardinality needs no exchange: every rank holds all rows
        assert shard.execute_query({"action": {"type": "Aggregated"}, "filterExpression": json.loads(query)["filterExpression"]}) == count
        # a leaf outside the window is reported, not silently wrong
        outside = shard_window[1] + 1 if rank < world - 1 else 1
        status, document = shard.execute_raw({"action": {"type": "Aggregated"},
                                              "filterExpression": {"type": "NucleotideEquals", "position": outside, "symbol": "A"}})
        assert status == 500 and "not resident" in document["message"]
        seen.extend(rows)
        shard.close()
    assert previous_end == model.positions
    assert seen == full


@pytest.mark.gpu
@pytest.mark.parametrize("launcher", ["self", "torchrun"])
def test_bench_two_rank_path_rehearsed_on_one_gpu(built, launcher):
    """bench.py for --gpus 2 both ways the driver may start it — `python bench.py --gpus 2` (bench.py spawns its ranks
    itself as child processes before touching the GPU) and `python -m torch.distributed.run ... bench.py --gpus 2` — one
    rank per process, position-range shards, the count table all-reduced through the engine's hook, except that both
    ranks share GPU 0 and the collective is gloo (--rehearse-on-one-gpu: RCCL refuses two ranks on one device).
    One JSON line, same rows."""
    root = os.path.dirname(HERE)
    options = ["--sequences", "200000", "--steps", "2", "--warmup", "1", "--no-also", "--no-cpu-baseline"]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    for name in ("RANK", "LOCAL_RANK", "WORLD_SIZE"):
        env.pop(name, None)
    single = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "1"] + options, capture_output=True, text=True, env=env, timeout=900)
    assert single.returncode == 0, single.stderr[-2000:]
    prefix = [sys.executable] if launcher == "self" else [
        sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", str(free_port())]
    double = subprocess.run(prefix + [os.path.join(root, "bench.py"), "--gpus", "2", "--rehearse-on-one-gpu"] + options,
                            capture_output=True, text=True, env=env, timeout=900)
    assert double.returncode == 0, double.stderr[-2000:]
    lines = double.stdout.strip().splitlines()
    assert len(lines) == 1, double.stdout[:500]
    a, b = json.loads(single.stdout.strip()), json.loads(lines[0])
    assert b["n_gpus"] == 2 and b["scaling"] == "strong" and b["config"]["sharding"].startswith("position-range x2")
    assert a["config"]["mutation_rows"] == b["config"]["mutation_rows"] > 0


@pytest.mark.gpu
def test_bench_two_rank_line_carries_every_leg_of_the_metric(built):
    """`bench.py --gpus 2` without --no-also (rehearsed on one GPU): beside the position-sharded headline the line carries the
    amino-acid leg on the position shards, BASELINE.json configs[4] on sequence-id shards (ONE batch of 200 queries, an
    all-reduce per count table) and the filter queries / s on those shards — the same answers as one rank gives."""
    root = os.path.dirname(HERE)
    options = ["--sequences", "200000", "--config4-sequences", "300000", "--steps", "2", "--warmup", "1", "--no-cpu-baseline"]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    for name in ("RANK", "LOCAL_RANK", "WORLD_SIZE"):
        env.pop(name, None)
    single = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "1", "--no-client-threads"] + options, capture_output=True, text=True, env=env,
                            timeout=1200)
    assert single.returncode == 0, single.stderr[-2000:]
    double = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--rehearse-on-one-gpu"] + options, capture_output=True, text=True,
                            env=env, timeout=1200)
    assert double.returncode == 0, double.stderr[-2000:]
    a, b = json.loads(single.stdout.strip().splitlines()[-1]), json.loads(double.stdout.strip().splitlines()[-1])
    assert a["config"]["mutation_rows"] == b["config"]["mutation_rows"] > 0
    assert a["also_amino_acid_full"]["mutation_rows"] == b["also_amino_acid_full"]["mutation_rows"] > 0 and "position-range x2" in b["also_amino_acid_full"]["workload"]
    assert b["also_config4"]["sharding"].startswith("sequence-id x2 (150000 rows per rank)") and b["also_config4"]["queries"] == 200
    assert b["also_config4"]["mutation_rows"] > 1000 and b["also_config4"]["value"] > 0
    assert b["filter_queries"]["count"] >= 0 and b["filter_queries"]["queries_per_s_1_client"] > 0


@pytest.mark.gpu
def test_native_comm_one_rank(built):
    """silo_gpu_comm_* / silo_gpu_allreduce_counts / silo_gpu_broadcast_bytes (include/silo_gpu.h) with a one-rank
    communicator, in a child process (RCCL is mapped there only): the sum over one rank and the broadcast from rank 0
    leave the buffer as it is, on a non-blocking stream, ordered with a memset before and a copy after."""
    code = r"""
import ctypes, sys
import numpy as np
sys.path[:0] = [@ROOT@, @PKG@]
from silo_amd import binding
lib = binding.load_library()
comm = binding.Comm(binding.comm_unique_id(), 0, 1, 0)
assert lib.silo_gpu_comm_rank(comm.handle) == 0 and lib.silo_gpu_comm_world(comm.handle) == 1
stream = ctypes.c_void_p()
binding._check(lib.silo_gpu_stream_create(ctypes.byref(stream)))
n = 29903 * 5
host = (np.arange(n, dtype=np.uint64) * 2654435761 % (1 << 32)).astype(np.uint32)
dev = ctypes.c_void_p()
binding._check(lib.silo_gpu_malloc(host.nbytes, ctypes.byref(dev)))
binding._check(lib.silo_gpu_memcpy_h2d(dev, host.ctypes.data_as(ctypes.c_void_p), host.nbytes, stream))
comm.all_reduce_counts(dev, n, stream)
comm.broadcast_bytes(dev, host.nbytes, 0, stream)
back = np.empty_like(host)
binding._check(lib.silo_gpu_memcpy_d2h(back.ctypes.data_as(ctypes.c_void_p), dev, back.nbytes, stream))
assert np.array_equal(back, host)
assert lib.silo_gpu_broadcast_bytes(comm.handle, dev, 4, 1, stream) < 0  # no such root
lib.silo_gpu_free(dev)
comm.close()
print("ok")
""".replace("@ROOT@", repr(os.path.dirname(HERE))).replace("@PKG@", repr(os.path.join(os.path.dirname(HERE), "lapis-silo_amd")))
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    done = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, timeout=600)
    assert done.returncode == 0 and done.stdout.strip().endswith("ok"), done.stderr[-2000:]
