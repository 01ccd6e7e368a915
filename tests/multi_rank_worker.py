"""Worker of tests/test_multi_rank.py: one rank of a gloo group (RANK / WORLD_SIZE / MASTER_* from the env).

mode "cpu":  host-side sharding logic only (no GPU): the engine's position windows + oracle counts per
             window, summed with a gloo all-reduce, must equal the unsharded oracle counts.
mode "gpu":  the real engine on cuda:0 in every rank (a 1-GPU box), sharded by position or by
             sequence id; the engine's all-reduce callback is backed by gloo through host memory.
             Rank 0 also runs every query through the CPU oracle on the unsharded data and reports the comparison.
mode "batch100": BASELINE.json configs[4] in shape — ONE batch of 100 queries (a lineage filter, every other one ANDed
             with a nucleotide predicate, each followed by Mutations or AminoAcidMutations) on sequence-id shards,
             against the oracle.
Prints one JSON line from rank 0.
"""
import ctypes
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "lapis-silo_amd")]

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

from oracle import dense  # noqa: E402
from oracle import silo_oracle as so  # noqa: E402
from oracle import synth as oracle_synth  # noqa: E402
from silo_amd import synth  # noqa: E402
from silo_amd.engine import Engine  # noqa: E402

N, P, L = 5000, 997, 60


def genomes():
    ref = synth.random_reference(P, "nuc", 3)
    gene = synth.random_reference(211, "aa", 4)
    doc = {
        "nucleotideSequences": [{"name": "main", "sequence": "".join("-ACGT"[s] for s in ref)}],
        "genes": [{"name": "S", "sequence": "".join("-ACDEFGHIKLMNPQRSTVWYBZ*X"[s] for s in gene)}],
    }
    return doc, ref, gene


def main():
    mode, shard = sys.argv[1], sys.argv[2]
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group("gloo", rank=rank, world_size=world)
    doc, ref, gene = genomes()
    tree = synth.make_lineage_tree(L)
    lineage = synth.assign_lineages(N, tree, 11)
    models = {
        "main": synth.make_model(N, ref, "nuc", tree, lineage, seed=5, store_index=0),
        "S": synth.make_model(N, gene, "aa", tree, lineage, seed=5, store_index=1),
    }
    for model in models.values():
        model.ambiguous_threshold = 1 << 13  # enough IUPAC codes for the sparse-leaf path
    member = tree.subtree(1)
    by_position = shard == "position"

    if mode == "batch100":
        by_position = False
    if mode == "cpu":
        engine = Engine(doc)
        engine.set_sharding(rank, world, True)
        out = {}
        for name, is_aa, symbols in (("main", False, [0, 1, 2, 3, 4]), ("S", True, list(range(21)) + [23])):
            begin, end = engine.position_window(name, is_aa)
            windows = [None] * world
            dist.all_gather_object(windows, (begin, end))
            sym = oracle_synth.symbol_matrix(models[name], np.arange(N), np.arange(begin, end))
            mask = member[lineage].astype(bool)
            table = np.zeros((models[name].positions, len(symbols)), dtype=np.int64)
            table[begin:end] = dense.mutation_counts(sym, mask, symbols)
            tensor = torch.from_numpy(table)
            dist.all_reduce(tensor)
            if rank == 0:
                full = oracle_synth.symbol_matrix(models[name], np.arange(N), np.arange(models[name].positions))
                want = dense.mutation_counts(full, mask, symbols)
                out[name] = {"windows": windows, "equal": bool(np.array_equal(tensor.numpy(), want))}
        if rank == 0:
            print(json.dumps(out), flush=True)
        dist.barrier()
        return

    # ---- gpu mode -------------------------------------------------------------------------------
    from silo_amd import binding

    lib = binding.load_library()

    # the collectives are handed the stream the engine's kernels run on (the request thread's own stream): the copies
    # below are enqueued on it and synchronise it, so they see the scans before and are seen by the kernels after
    def all_reduce(device_ptr, n, stream):
        host = np.empty(n, dtype=np.int32)
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

    engine = Engine(doc)
    engine.set_sharding(rank, world, by_position, all_reduce)
    if by_position:
        engine.set_broadcast(broadcast)
    if by_position:
        rows = slice(0, N)
    else:
        rows = slice(N * rank // world, N * (rank + 1) // world)
    n_local = rows.stop - rows.start
    part = engine.add_partition(n_local)
    for name, is_aa in (("main", False), ("S", True)):
        begin, end = engine.position_window(name, is_aa)
        sym = oracle_synth.symbol_matrix(models[name], np.arange(rows.start, rows.stop), np.arange(begin, end))
        chars = np.frombuffer(b"-ACGTRYSWKMBDHVN" if not is_aa else b"-ACDEFGHIKLMNPQRSTVWYBZ*X", dtype=np.uint8)[sym]
        engine.append_sequences(part, name, is_aa, 0, [bytes(row).decode("latin-1") for row in chars])
    engine.set_lineage_column_ids(part, "pango_lineage", tree.names, lineage[rows])
    # metadata columns: every rank holds the values of its rows (all rows under position sharding)
    metadata_rng = np.random.default_rng(3)
    country = metadata_rng.integers(0, 7, size=N)
    age = metadata_rng.integers(0, 90, size=N)
    engine.append_metadata(part, "country", "indexed_string", [f"C{c}" for c in country[rows]])
    engine.append_metadata(part, "age", "int", [str(a) if a % 11 else "" for a in age[rows]])
    engine.finalize()

    oracle_db = None
    if rank == 0:  # the unsharded database in the CPU oracle (row and position order = the engine's global order)
        oracle_db = so.Database({"main": list(ref)}, {"S": list(gene)})
        oracle_db.set_config([("country", "indexed_string"), ("age", "int")], "country")
        chars = {}
        for name, is_aa in (("main", False), ("S", True)):
            sym = oracle_synth.symbol_matrix(models[name], np.arange(N), np.arange(models[name].positions))
            lut = np.frombuffer(b"-ACGTRYSWKMBDHVN" if not is_aa else b"-ACDEFGHIKLMNPQRSTVWYBZ*X", dtype=np.uint8)
            chars[name] = [bytes(row).decode("latin-1") for row in lut[sym]]
        oracle_partition = oracle_db.add_partition({"main": chars["main"]}, {"S": chars["S"]}, [tree.names[i] for i in lineage])
        oracle_db.add_metadata(oracle_partition, [{"country": f"C{c}", "age": str(a) if a % 11 else ""} for c, a in zip(country, age)])

    def oracle_answers(queries):
        answers = []
        for query in queries:
            try:
                answers.append([200, json.loads(json.dumps({"queryResult": so.execute_query(oracle_db, query)}))])
            except so.QueryParseException as error:
                answers.append([400, {"error": "Bad request", "message": str(error)}])
        return answers

    if mode == "batch100":
        sizes = np.bincount(lineage, minlength=L)
        names = [tree.names[i] for i in np.argsort(-sizes, kind="stable")[:50]]
        queries = []
        for k in range(100):
            expression = {"type": "PangoLineage", "column": "pango_lineage", "value": names[k % 50], "includeSublineages": k % 3 != 0}
            if k % 2 == 1:
                predicate = {"type": "NucleotideEquals", "position": 1 + (37 * k) % P, "symbol": "ACGT-"[k % 5]}
                expression = {"type": "And", "children": [expression, {"type": "Not", "child": predicate} if k % 4 == 1 else {"type": "Maybe", "child": predicate}]}
            action = {"type": "Mutations", "minProportion": 0.05} if k % 4 < 2 else {"type": "AminoAcidMutations", "minProportion": 0.05}
            queries.append({"action": action, "filterExpression": expression})
        batched = engine.execute_batch(queries)
        dist.barrier()
        if rank == 0:
            want = oracle_answers(queries)
            print(json.dumps({
                "queries": len(queries), "equal": [got == exp for got, exp in zip(json.loads(json.dumps(batched)), want)],
                "rows": [len(document.get("queryResult", [])) for _, document in batched],
            }), flush=True)
        engine.close()
        return

    queries = [
        {"action": {"type": "Mutations", "minProportion": 0.02}, "filterExpression": {"type": "PangoLineage", "column": "pango_lineage", "value": "B.1", "includeSublineages": True}},
        {"action": {"type": "AminoAcidMutations", "minProportion": 0.0}, "filterExpression": {"type": "True"}},
        {"action": {"type": "Mutations", "minProportion": 0.5}, "filterExpression": {"type": "PangoLineage", "column": "pango_lineage", "value": "B.2.3", "includeSublineages": False}},
        {"action": {"type": "Aggregated"}, "filterExpression": {"type": "PangoLineage", "column": "pango_lineage", "value": "B.1", "includeSublineages": True}},
        # filter leaves spread over the genome: under position sharding they are fetched from their owner rank
        {"action": {"type": "Mutations", "minProportion": 0.3}, "filterExpression": {"type": "And", "children": [
            {"type": "Not", "child": {"type": "NucleotideEquals", "position": 10, "symbol": "-"}},
            {"type": "N-Of", "numberOfMatchers": 1, "matchExactly": False, "children": [
                {"type": "HasNucleotideMutation", "position": 900}, {"type": "NucleotideEquals", "position": 500, "symbol": "N"},
                {"type": "Maybe", "child": {"type": "NucleotideEquals", "position": 333, "symbol": "."}}]},
            {"type": "Not", "child": {"type": "AminoAcidEquals", "sequenceName": "S", "position": 200, "symbol": "X"}}]}},
        {"action": {"type": "Aggregated"}, "filterExpression": {"type": "Or", "children": [
            {"type": "NucleotideEquals", "position": 997, "symbol": "R"}, {"type": "NucleotideEquals", "position": 1, "symbol": "Y"},
            {"type": "HasAminoAcidMutation", "sequenceName": "S", "position": 100}]}},
        # metadata predicates feed the same fused filter program on every rank
        {"action": {"type": "Aggregated"}, "filterExpression": {"type": "And", "children": [
            {"type": "StringEquals", "column": "country", "value": "C3"}, {"type": "IntBetween", "column": "age", "from": 30, "to": None},
            {"type": "Not", "child": {"type": "NucleotideEquals", "position": 700, "symbol": "-"}}]}},
        {"action": {"type": "Mutations", "minProportion": 0.1}, "filterExpression": {"type": "IntEquals", "column": "age", "value": None}},
    ]
    if by_position:  # row-wise actions need every row on the rank
        queries.append({"action": {"type": "Aggregated", "groupByFields": ["country"], "orderByFields": ["country"]},
                        "filterExpression": {"type": "NucleotideEquals", "position": 950, "symbol": "A"}})
        queries.append({"action": {"type": "Details", "fields": ["age", "country"], "orderByFields": ["age", "country"], "limit": 7},
                        "filterExpression": {"type": "NucleotideEquals", "position": 20, "symbol": "T"}})
    results = [engine.execute_raw(q) for q in queries]
    # the same queries as one batch: scans share plane passes, the count tables are reduced after the launches
    batched = engine.execute_batch(queries)
    if batched != results:
        raise AssertionError(f"rank {rank}: batched results differ from one-by-one results")
    # ranks that do NOT run the same query: their collectives pair up all the same, the query fingerprints carried by the
    # all-reduce give it away and every rank answers 500 instead of a mix of two queries' counts
    def differing(action):
        return {"action": action, "filterExpression": {"type": "PangoLineage", "column": "pango_lineage", "value": f"B.{1 + rank % 2}", "includeSublineages": True}}

    mismatched = [engine.execute_raw(differing({"type": "Mutations", "minProportion": 0.02}))]
    if not by_position:  # Aggregated has a collective only where the ranks hold different rows
        mismatched.append(engine.execute_raw(differing({"type": "Aggregated"})))
    mismatched += engine.execute_batch([differing({"type": "Mutations", "minProportion": 0.1}), queries[0]])[:1]
    refused = [status == 500 and "did not run the same query" in document.get("message", "") for status, document in mismatched]
    after = engine.execute_raw(queries[0])  # and the engine is fine afterwards
    gathered = [None] * world
    dist.all_gather_object(gathered, refused + [after == results[0]])
    dist.barrier()
    if rank == 0:
        want = oracle_answers(queries)
        got = json.loads(json.dumps(results))
        print(json.dumps({"results": results, "matches_oracle": [g == w for g, w in zip(got, want)], "mismatch_refused": gathered}), flush=True)
    engine.close()


if __name__ == "__main__":
    main()
