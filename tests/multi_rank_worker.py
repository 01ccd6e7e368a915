"""Worker of tests/test_multi_rank.py: one rank of a gloo group (RANK / WORLD_SIZE / MASTER_* from the env).

mode "cpu":  host-side sharding logic only (no GPU): the engine's position windows + oracle counts per
             window, summed with a gloo all-reduce, must equal the unsharded oracle counts.
mode "gpu":  the real engine on cuda:0 in every rank (a 1-GPU box), sharded by position or by
             sequence id; the engine's all-reduce callback is backed by gloo through host memory.
             Rank 0 also runs every query through the CPU oracle on the unsharded data and reports the comparison.
mode "batch100": BASELINE.json configs[4] in shape — ONE batch of 100 queries (a lineage filter, every other one ANDed
             with a nucleotide predicate, each followed by Mutations or AminoAcidMutations) on sequence-id shards,
             against the oracle.
mode "gpu_big": 140 000 rows (>= 65 536 on every shard: finalize re-encodes the stores and the tiled kernels, the escape pass
             and the derived-symbol passes run), filled by the device generator; rank 0 checks every answer against the naive
             counter (oracle/dense.py) over the generator's CPU twin — silo_oracle.py needs minutes per query at this size.
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

# rows of the whole database: 5 000 by default (short rows: identity planes, the row-wave kernel); SILO_TEST_ROWS >= 140 000 puts
# >= 65 536 rows on every shard, where finalize re-encodes the stores (derived symbols, one-hot rows, slice-major escape keys,
# runs of the missing symbol) and the tiled kernels run — sharding + collective + adaptive layout together against the oracle
N = int(os.environ.get("SILO_TEST_ROWS", "5000"))
P, L = 997, 60


def genomes():
    ref = synth.random_reference(P, "nuc", 3)
    gene = synth.random_reference(211, "aa", 4)
    doc = {
        "nucleotideSequences": [{"name": "main", "sequence": "".join("-ACGT"[s] for s in ref)}],
        "genes": [{"name": "S", "sequence": "".join("-ACDEFGHIKLMNPQRSTVWYBZ*X"[s] for s in gene)}],
    }
    return doc, ref, gene


NUC_CHARS = "-ACGTRYSWKMBDHVN"
AA_CHARS = "-ACDEFGHIKLMNPQRSTVWYBZ*X"
# what a symbol may be under Maybe: itself or an ambiguity code that may stand for it (nucleotide_symbol_equals.cpp:28-73)
NUC_UPPER = {0: [0], 1: [1, 5, 10, 8, 12, 13, 14, 15], 2: [2, 6, 10, 7, 11, 13, 14, 15], 3: [3, 5, 9, 7, 11, 12, 14, 15], 4: [4, 6, 9, 8, 11, 12, 13, 15]}


def dense_mutation_rows(symbols, mask, reference, valid_symbols, chars, min_proportion, sequence_name):
    """The rows Mutations / AminoAcidMutations must give, by direct counting: mutations.cpp:184-232 over oracle/dense.py's table."""
    table = dense.mutation_counts(symbols, mask, valid_symbols).astype(np.int64)
    rows = []
    for position in range(table.shape[0]):
        total = int(table[position].sum())
        if total == 0:
            continue
        must_exceed = 0 if min_proportion == 0 else int(np.uint32(np.ceil(np.float64(total) * min_proportion) - 1))
        for k, symbol in enumerate(valid_symbols):
            count = int(table[position][k])
            if symbol != reference[position] and count > must_exceed:
                rows.append({"count": count, "mutation": f"{chars[reference[position]]}{position + 1}{chars[symbol]}", "proportion": count / total,
                             "sequenceName": sequence_name})
    return rows


def big_case(rank, world, by_position):
    """140 000 rows on two shards, filled by the device generator; rank 0 checks against the naive counter."""
    from silo_amd import alphabet, binding

    n, p_nuc, p_aa, n_lineages = 140_000, 301, 97, 60
    ref = synth.random_reference(p_nuc, "nuc", 3)
    gene = synth.random_reference(p_aa, "aa", 4)
    doc = {"nucleotideSequences": [{"name": "main", "sequence": "".join(NUC_CHARS[s] for s in ref)}],
           "genes": [{"name": "S", "sequence": "".join(AA_CHARS[s] for s in gene)}]}
    tree = synth.make_lineage_tree(n_lineages)
    lineage = synth.assign_lineages(n, tree, 11)
    lib = binding.load_library()

    def all_reduce(device_ptr, count, stream):
        host = np.empty(count, dtype=np.int32)
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

    # the rows of a shard: all of them under position sharding; a slice, generated from the shard's own seed, by sequence id
    shards = [slice(0, n)] if by_position else [slice(n * r // world, n * (r + 1) // world) for r in range(world)]

    def models_of(shard_index):
        rows = shards[shard_index]
        seed = 5 + 1000 * shard_index
        made = {"main": synth.make_model(rows.stop - rows.start, ref, "nuc", tree, lineage[rows], seed=seed, store_index=0, table_seed=5),
                "S": synth.make_model(rows.stop - rows.start, gene, "aa", tree, lineage[rows], seed=seed, store_index=1, table_seed=5)}
        for model in made.values():
            model.ambiguous_threshold = 1 << 11  # ~1e-4 of the cells: ambiguity codes at every position
            # the model gives every other sequence a run of the missing symbol of up to half the store: on these short stores that is
            # no alignment (the runs would not even be smaller than a quarter of the symbol's plane, and the plane would stay): one in ten
            model.missing_len[np.random.default_rng(seed).random(len(model.missing_len)) < 0.8] = 0
        return made

    mine = models_of(0 if by_position else rank)
    engine = Engine(doc)
    engine.set_sharding(rank, world, by_position, all_reduce)
    if by_position:
        engine.set_broadcast(broadcast)
    n_local = shards[0 if by_position else rank].stop - shards[0 if by_position else rank].start
    part = engine.add_partition(n_local)
    for name, is_aa in (("main", False), ("S", True)):
        engine.generate_synthetic(part, name, is_aa, mine[name], engine.position_window(name, is_aa))
    engine.set_lineage_column_ids(part, "pango_lineage", tree.names, lineage[shards[0 if by_position else rank]])
    engine.finalize()
    store = engine.partition_store(0)
    window = engine.position_window("main", False)
    layout = {"most_common_rows_per_position": int(lib.silo_gpu_store_scan_planes(store.handle, 0)), "runs": int(lib.silo_gpu_store_scan_runs(store.handle, 0)),
              "plane_rows": int(lib.silo_gpu_store_scan_rows(store.handle, 0, 0, window[1] - window[0])), "positions": window[1] - window[0],
              "escape_keys": int(lib.silo_gpu_store_scan_escapes(store.handle, 0))}
    # re-encoded with derived symbols: the runs of the missing symbol are part of the scan, fewer plane rows than 3 identity planes per position
    re_encoded = layout["runs"] > 0 and layout["plane_rows"] < 3 * layout["positions"] and layout["escape_keys"] > 0

    def lineage_filter(name, sublineages=True):
        return {"type": "PangoLineage", "column": "pango_lineage", "value": name, "includeSublineages": sublineages}

    reference_33 = NUC_CHARS[ref[32]]
    sizes = np.bincount(lineage, minlength=n_lineages)
    frequent = [tree.names[i] for i in np.argsort(-sizes, kind="stable")[:16]]
    queries = [
        {"action": {"type": "Mutations", "minProportion": 0.02}, "filterExpression": lineage_filter("B.1")},
        {"action": {"type": "AminoAcidMutations", "minProportion": 0.0}, "filterExpression": {"type": "True"}},
        {"action": {"type": "Mutations", "minProportion": 0.0}, "filterExpression": {"type": "True"}},
        {"action": {"type": "Mutations", "minProportion": 0.5}, "filterExpression": lineage_filter("B.2.3", False)},
        {"action": {"type": "AminoAcidMutations", "minProportion": 0.05}, "filterExpression": lineage_filter("B.2")},
        {"action": {"type": "Mutations", "minProportion": 0.3}, "filterExpression": {"type": "And", "children": [
            {"type": "Not", "child": {"type": "NucleotideEquals", "position": 10, "symbol": "-"}},
            {"type": "Maybe", "child": {"type": "NucleotideEquals", "position": 33, "symbol": reference_33}}]}},
        {"action": {"type": "Aggregated"}, "filterExpression": lineage_filter("B.1")},
        {"action": {"type": "Aggregated"}, "filterExpression": {"type": "NucleotideEquals", "position": 150, "symbol": NUC_CHARS[ref[149]]}},
        {"action": {"type": "Aggregated"}, "filterExpression": {"type": "NucleotideEquals", "position": 200, "symbol": "N"}},
        {"action": {"type": "Aggregated"}, "filterExpression": {"type": "Or", "children": [
            {"type": "NucleotideEquals", "position": 301, "symbol": "R"}, {"type": "NucleotideEquals", "position": 1, "symbol": "-"}]}},
    ]
    queries += [{"action": {"type": "Mutations", "minProportion": 0.05}, "filterExpression": lineage_filter(name)} for name in frequent[:6]]
    queries += [{"action": {"type": "AminoAcidMutations", "minProportion": 0.05}, "filterExpression": lineage_filter(name, False)} for name in frequent[:6]]
    results = [engine.execute_raw(q) for q in queries]
    batch = [{"action": {"type": "Mutations" if k % 2 == 0 else "AminoAcidMutations", "minProportion": 0.04}, "filterExpression": lineage_filter(frequent[k], k % 3 != 0)}
             for k in range(16)]
    batched = engine.execute_batch(batch)
    dist.barrier()
    if rank == 0:
        # the whole database as characters' symbols, from the generator's CPU twin: shard by shard (sequence ids) or whole
        symbols = {}
        for name, positions in (("main", p_nuc), ("S", p_aa)):
            symbols[name] = np.vstack([
                oracle_synth.symbol_matrix(models_of(k)[name] if k > 0 or not by_position else mine[name], np.arange(shards[k].stop - shards[k].start), np.arange(positions))
                for k in range(len(shards))])
        valid_nuc, valid_aa = list(alphabet.NUCLEOTIDE.valid_mutation_symbols), list(alphabet.AMINO_ACID.valid_mutation_symbols)

        def mask_of(expression):
            kind = expression["type"]
            if kind == "True":
                return np.ones(n, dtype=bool)
            if kind == "PangoLineage":
                index = tree.names.index(expression["value"])
                return tree.subtree(index)[lineage].astype(bool) if expression["includeSublineages"] else lineage == index
            if kind == "NucleotideEquals":
                return symbols["main"][:, expression["position"] - 1] == NUC_CHARS.index(expression["symbol"])
            if kind == "Maybe":  # of a NucleotideEquals
                child = expression["child"]
                return np.isin(symbols["main"][:, child["position"] - 1], NUC_UPPER[NUC_CHARS.index(child["symbol"])])
            if kind == "Not":
                return ~mask_of(expression["child"])
            if kind in ("And", "Or"):
                parts = [mask_of(child) for child in expression["children"]]
                return np.logical_and.reduce(parts) if kind == "And" else np.logical_or.reduce(parts)
            raise ValueError(kind)

        def expected(query):
            mask = mask_of(query["filterExpression"])
            action = query["action"]
            if action["type"] == "Aggregated":
                return [{"count": int(mask.sum())}]
            if action["type"] == "Mutations":
                return dense_mutation_rows(symbols["main"], mask, ref, valid_nuc, NUC_CHARS, action["minProportion"], "main")
            return dense_mutation_rows(symbols["S"], mask, gene, valid_aa, AA_CHARS, action["minProportion"], "S")

        equal = [status == 200 and document["queryResult"] == expected(query) for (status, document), query in zip(results, queries)]
        batch_equal = [status == 200 and document["queryResult"] == expected(query) for (status, document), query in zip(batched, batch)]
        print(json.dumps({"rows_per_rank": n_local, "re_encoded": bool(re_encoded), "layout": layout, "equal": equal, "batch_equal": batch_equal,
                          "nonempty": sum(1 for _, document in results if len(document.get("queryResult", [])) > 0)}), flush=True)
    engine.close()


def main():
    mode, shard = sys.argv[1], sys.argv[2]
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group("gloo", rank=rank, world_size=world)
    if mode == "gpu_big":
        big_case(rank, world, shard == "position")
        return
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
    engine.set_schema("country")  # (the oracle's primary key below: FastaAligned labels its rows with it)
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
    concatenated = None
    if not by_position:
        # row-wise actions on sequence-id shards: every rank answers for ITS rows, the front end concatenates (no collective)
        row_queries = [
            {"action": {"type": "Details", "fields": ["age", "country"]}, "filterExpression": {"type": "NucleotideEquals", "position": 20, "symbol": "T"}},
            {"action": {"type": "FastaAligned", "sequenceName": "main"}, "filterExpression": {"type": "And", "children": [
                {"type": "PangoLineage", "column": "pango_lineage", "value": "B.2.3", "includeSublineages": False},
                {"type": "NucleotideEquals", "position": 30, "symbol": "-"}]}},
        ]
        local = [engine.execute_raw(q) for q in row_queries]
        everyones = [None] * world
        dist.all_gather_object(everyones, local)
        if rank == 0:
            concatenated = []
            for k, query in enumerate(row_queries):
                assert all(answers[k][0] == 200 for answers in everyones), [answers[k] for answers in everyones]
                rows = [row for answers in everyones for row in answers[k][1]["queryResult"]]
                want = so.execute_query(oracle_db, query)
                key = lambda row: json.dumps(row, sort_keys=True)
                concatenated.append(sorted(map(key, rows)) == sorted(map(key, json.loads(json.dumps(want)))) and len(rows) > 0)
        # what does merge rows across shards is refused
        status, document = engine.execute_raw({"action": {"type": "Aggregated", "groupByFields": ["country"]}, "filterExpression": {"type": "True"}})
        assert world == 1 or (status == 500 and "sharded by sequence id" in document["message"]), (status, document)
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
        print(json.dumps({"results": results if N <= 20000 else [[status, len(json.dumps(document))] for status, document in results],
                          "matches_oracle": [g == w for g, w in zip(got, want)], "mismatch_refused": gathered, "row_actions_concatenate": concatenated,
                          "rows_per_rank": n_local}), flush=True)
    engine.close()


if __name__ == "__main__":
    main()
