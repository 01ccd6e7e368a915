"""Oracle-checked device runs of the BASELINE.json configurations the benchmark measures.

configs[2]  the exact 32-leaf tree of bench.filter_query — And(Or(8), 3-of-8, Not(Or(8)), Maybe(And(8))) — on a synthetic
            store the oracle can hold (count AND sequence-id set), and at the full 10 M sequences against a numpy
            evaluation of the downloaded leaf planes (the planes themselves against the CPU twin of the generator).
configs[3]  the 10 M-sequence Mutations scan: the compact-index scan (K1i) against the scan of the full code planes over
            the whole genome, and both against the C port of the reference algorithm (oracle/roaring_port.c) on a sample
            of positions — the check bench.py's cpu_baseline leg makes, as a test.
(configs[3] / [4] across ranks: tests/test_multi_rank.py.)
"""
import ctypes
import json

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

import bench  # noqa: E402
from oracle import cpu_port, dense  # noqa: E402
from oracle import silo_oracle as so  # noqa: E402
from oracle import synth as oracle_synth  # noqa: E402
from silo_amd import binding, synth  # noqa: E402
from silo_amd.engine import Engine  # noqa: E402

NUC_CHARS = "-ACGTRYSWKMBDHVN"


def bits_of(words):
    """uint64 words (bit i of word w = row 64 w + i) -> Python int with bit r = row r, the oracle's bitmap form."""
    return int.from_bytes(np.ascontiguousarray(words, dtype="<u8").tobytes(), "little")


# ---- configs[2] on a store the oracle can hold ----------------------------------------------------------------------
@pytest.fixture(scope="module")
def small_store(built):
    n, positions = 20_000, 1_500
    tree = synth.make_lineage_tree(200)
    lineage = synth.assign_lineages(n, tree, 7)
    reference = synth.random_reference(positions, "nuc", 11)
    model = synth.make_model(n, reference, "nuc", tree, lineage, seed=13, store_index=0)
    model.ambiguous_threshold = 1 << 12  # IUPAC codes often enough that Maybe meets sparsely stored planes
    genomes = {"nucleotideSequences": [{"name": "main", "sequence": "".join(NUC_CHARS[s] for s in reference)}], "genes": []}
    engine = Engine(genomes)
    partition = engine.add_partition(n)
    engine.generate_synthetic(partition, "main", False, model)
    engine.set_lineage_column_ids(partition, "pango_lineage", tree.names, lineage)
    engine.finalize()
    symbols = oracle_synth.symbol_matrix(model, np.arange(n), np.arange(positions))
    lut = np.frombuffer(NUC_CHARS.encode(), dtype=np.uint8)
    database = so.Database({"main": list(reference)}, {})
    database.add_partition({"main": [bytes(row).decode("latin-1") for row in lut[symbols]]}, {}, [tree.names[i] for i in lineage])
    yield engine, database, model, tree, symbols
    engine.close()


def config2_variants(model, tree):
    """The benchmark's tree and the rewrites around it that the reference's compile() treats differently."""
    tree_query = json.loads(bench.filter_query(model, tree))
    expression = tree_query["filterExpression"]
    or8, nof, not_or8, maybe_and8 = expression["children"]
    leaves = maybe_and8["child"]["children"] + or8["children"] + nof["children"] + not_or8["child"]["children"]
    assert len(leaves) == 32 and len({(leaf["position"], leaf["symbol"]) for leaf in leaves}) == 32
    variants = {"configs[2] tree": expression, "Or(8)": or8, "3-of-8": nof, "Not(Or(8))": not_or8, "Maybe(And(8))": maybe_and8}
    variants["exactly 3-of-8"] = dict(nof, matchExactly=True)
    variants["Maybe(exactly 2-of-8)"] = {"type": "Maybe", "child": dict(nof, numberOfMatchers=2, matchExactly=True)}  # nof.cpp:220-258
    variants["Not(tree)"] = {"type": "Not", "child": expression}
    variants["Maybe(tree)"] = {"type": "Maybe", "child": expression}
    variants["Exact(Maybe(And(8)))"] = {"type": "Exact", "child": maybe_and8}
    variants["5-of-32 over all leaves"] = {"type": "N-Of", "numberOfMatchers": 5, "matchExactly": False, "children": leaves}
    variants["Or(32)"] = {"type": "Or", "children": leaves}
    variants["Maybe(Or(32))"] = {"type": "Maybe", "child": {"type": "Or", "children": leaves}}
    variants["And(Not x 8)"] = {"type": "And", "children": [{"type": "Not", "child": leaf} for leaf in leaves[:8]]}
    return variants


def test_config2_tree_count_and_id_set_match_the_oracle(small_store):
    engine, database, model, tree, _ = small_store
    n = model.n_sequences
    counts = {}
    for name, expression in config2_variants(model, tree).items():
        query = {"action": {"type": "Aggregated"}, "filterExpression": expression}
        want_rows = so.execute_query(database, query)
        assert engine.execute_query(query) == want_rows, name
        want_bits = so.evaluate_filters(database, so.parse_expression(expression))[0] & ((1 << n) - 1)
        words, cardinality = engine.evaluate_filter(expression)
        assert bits_of(words) == want_bits, name  # the sequence-id set, bit for bit
        assert cardinality == want_rows[0]["count"] == bin(want_bits).count("1"), name
        counts[name] = cardinality
    # the check is not vacuous: the tree selects some rows and not all, and its parts differ
    assert 0 < counts["configs[2] tree"] < counts["Or(8)"] < n and 0 < counts["3-of-8"] < n and counts["Maybe(And(8))"] > 0
    # the same queries as one batch (the multi-query filter launch) give the same documents
    queries = [{"action": {"type": "Aggregated"}, "filterExpression": e} for e in config2_variants(model, tree).values()]
    assert engine.execute_batch(queries) == [engine.execute_raw(q) for q in queries]


def test_config2_tree_under_mutations_matches_the_oracle(small_store):
    engine, database, model, tree, _ = small_store
    expression = json.loads(bench.filter_query(model, tree))["filterExpression"]
    for action in ({"type": "Mutations", "minProportion": 0.02}, {"type": "Mutations", "minProportion": 0.0, "orderByFields": ["mutation"], "limit": 300}):
        query = {"action": action, "filterExpression": expression}
        assert engine.execute_query(query) == json.loads(json.dumps(so.execute_query(database, query)))


# ---- the full-size store: configs[2] and configs[3] at 10 M sequences ----------------------------------------------
FULL_N = 10_000_000


@pytest.fixture(scope="module")
def full_store(built):
    engine, model, tree, lineage, window = bench.build_engine(FULL_N, 0, 1, None, 0)
    assert window == (0, model.positions)
    yield engine, model, tree, lineage
    engine.close()


def leaf_plane(engine, expression):
    words, cardinality = engine.evaluate_filter(expression)
    assert int(np.unpackbits(words.view(np.uint8)).sum()) == cardinality
    return words


def check_leaf_planes_against_the_generator(model, all_leaves, plain, upper):
    """24 plain leaf planes + 8 planes under Maybe against the CPU twin of the generator (oracle/synth.py) on 60 000 sampled rows."""
    rng = np.random.default_rng(5)
    rows = np.unique(rng.integers(0, FULL_N, size=60_000))
    ambiguity = {s: set(codes) for s, codes in enumerate([[0], [1, 5, 10, 8, 12, 13, 14, 15], [2, 6, 10, 7, 11, 13, 14, 15], [3, 5, 9, 7, 11, 12, 14, 15],
                                                         [4, 6, 9, 8, 11, 12, 13, 15]])}  # nucleotide_symbol_equals.cpp:28-73
    positions = np.array([leaf["position"] - 1 for leaf in all_leaves])
    symbols = oracle_synth.symbol_matrix(model, rows, positions)  # [rows][32]
    for k, leaf in enumerate(all_leaves):
        symbol = NUC_CHARS.index(leaf["symbol"])
        plane = plain[k] if k < 24 else upper[k - 24]
        got = (plane[rows >> 6] >> (rows & 63).astype(np.uint64)) & np.uint64(1)
        accepted = {symbol} if k < 24 else ambiguity[symbol]
        want = np.isin(symbols[:, k], list(accepted))
        assert np.array_equal(got.astype(bool), want), leaf


def test_config2_tree_at_10m_equals_numpy_over_its_leaf_planes(full_store):
    engine, model, tree, lineage = full_store
    expression = json.loads(bench.filter_query(model, tree))["filterExpression"]
    or8, nof, not_or8, maybe_and8 = expression["children"]
    plain = [leaf_plane(engine, leaf) for leaf in or8["children"] + nof["children"] + not_or8["child"]["children"]]
    upper = [leaf_plane(engine, {"type": "Maybe", "child": leaf}) for leaf in maybe_and8["child"]["children"]]

    # the leaf planes against the CPU twin of the generator, on a sample of rows
    all_leaves = or8["children"] + nof["children"] + not_or8["child"]["children"] + maybe_and8["child"]["children"]
    check_leaf_planes_against_the_generator(model, all_leaves, plain, upper)

    # the tree over the planes, in numpy
    valid = np.zeros(len(plain[0]), dtype=np.uint64)
    valid[: FULL_N // 64] = ~np.uint64(0)
    if FULL_N % 64:
        valid[FULL_N // 64] = np.uint64((1 << (FULL_N % 64)) - 1)
    any_of = np.bitwise_or.reduce(plain[0:8])
    matches = np.zeros(FULL_N + 64, dtype=np.uint8)[: len(valid) * 64]
    for plane in plain[8:16]:
        matches += np.unpackbits(plane.view(np.uint8), bitorder="little")
    at_least_3 = np.packbits(matches >= 3, bitorder="little").view(np.uint64)
    none_of = ~np.bitwise_or.reduce(plain[16:24]) & valid
    all_maybe = np.bitwise_and.reduce(upper)
    want = any_of & at_least_3 & none_of & all_maybe
    words, cardinality = engine.evaluate_filter(expression)
    assert np.array_equal(words, want)
    want_count = int(np.unpackbits(want.view(np.uint8)).sum())
    assert cardinality == want_count
    query = {"action": {"type": "Aggregated"}, "filterExpression": expression}
    assert engine.execute_query(query) == [{"count": want_count}]
    assert 0 < want_count < FULL_N
    # many of them in one batch (one multi-query launch): every answer the same count
    assert engine.execute_batch([query] * 70) == [(200, {"queryResult": [{"count": want_count}]})] * 70


def test_config2_batch_of_64_different_trees_at_10m_equals_numpy(full_store):
    """The batch that carries BASELINE's second metric (bench.py filter_workload): 64 trees of the config-2 shape over 64 x 32
    DIFFERENT leaves, as one silo_engine_execute_batch (one k_filter_eval_batch launch).  Every count is checked against the
    tree evaluated in numpy over the downloaded leaf planes (variants > 0 ask for the reference symbol under N-Of and
    Maybe(And): planes of DERIVED symbols, rebuilt as "no other symbol and not missing"), and most of them select rows."""
    engine, model, tree, lineage = full_store
    valid = None
    want_counts = []
    queries = []
    for variant in range(64):
        text = bench.filter_query(model, tree, variant)
        queries.append(json.loads(text))
        or8, nof, not_or8, maybe_and8 = queries[-1]["filterExpression"]["children"]
        plain = [leaf_plane(engine, leaf) for leaf in or8["children"] + nof["children"] + not_or8["child"]["children"]]
        upper = [leaf_plane(engine, {"type": "Maybe", "child": leaf}) for leaf in maybe_and8["child"]["children"]]
        if variant in (1, 40):  # planes of derived symbols (the reference symbol of a position) against the generator's twin
            check_leaf_planes_against_the_generator(model, or8["children"] + nof["children"] + not_or8["child"]["children"] + maybe_and8["child"]["children"], plain, upper)
        if valid is None:
            valid = np.zeros(len(plain[0]), dtype=np.uint64)
            valid[: FULL_N // 64] = ~np.uint64(0)
            if FULL_N % 64:
                valid[FULL_N // 64] = np.uint64((1 << (FULL_N % 64)) - 1)
        any_of = np.bitwise_or.reduce(plain[0:8])
        # at least 3 of 8, with a bit-sliced counter over the words (no per-row array)
        ones = np.zeros_like(valid)
        twos = np.zeros_like(valid)
        fours = np.zeros_like(valid)
        for plane in plain[8:16]:
            carry = ones & plane
            ones ^= plane
            carry2 = twos & carry
            twos ^= carry
            fours |= carry2  # saturates at "4 or more"
        at_least_3 = fours | (twos & ones)
        none_of = ~np.bitwise_or.reduce(plain[16:24]) & valid
        all_maybe = np.bitwise_and.reduce(upper)
        want = any_of & at_least_3 & none_of & all_maybe
        want_counts.append(int(np.unpackbits(want.view(np.uint8)).sum()))
    got = engine.execute_batch(queries)
    assert got == [(200, {"queryResult": [{"count": count}]}) for count in want_counts]
    assert sum(1 for count in want_counts if count > 0) >= 32, want_counts
    assert [engine.execute_query(queries[k]) for k in (1, 17, 63)] == [[{"count": want_counts[k]}] for k in (1, 17, 63)]


def scan_table(lib, store, filt, begin, end, seqstore_id=0, n_symbols=5):
    n = (end - begin) * n_symbols
    device = ctypes.c_void_p()
    binding._check(lib.silo_gpu_malloc(4 * n, ctypes.byref(device)))
    binding._check(lib.silo_gpu_memset_async(device, 0, 4 * n, None))
    binding._check(lib.silo_gpu_mutations_scan(store.handle, seqstore_id, filt, begin, end, device, None))
    table = np.empty(n, dtype=np.uint32)
    binding._check(lib.silo_gpu_memcpy_d2h(table.ctypes.data_as(ctypes.c_void_p), device, table.nbytes, None))
    lib.silo_gpu_free(device)
    return table.reshape(end - begin, n_symbols)


def test_config3_scan_at_10m_adaptive_planes_identity_planes_and_c_port_agree(full_store):
    engine, model, tree, lineage = full_store
    lib = binding.load_library()
    store = engine.partition_store(0)
    member = tree.subtree(tree.names.index(bench.QUERY_LINEAGE))
    positions = model.positions

    def lineage_filter(handle):
        filt = ctypes.c_void_p()
        binding._check(lib.silo_gpu_bitset_alloc(handle, ctypes.byref(filt)))
        binding._check(lib.silo_gpu_bitset_from_lineages(handle, filt, member.ctypes.data_as(ctypes.c_void_p), len(member), None))
        return filt

    filt = lineage_filter(store.handle)
    try:
        assert lib.silo_gpu_store_scan_planes(store.handle, 0) == 0, "the 10 M nucleotide store is expected to derive the most numerous symbol of its positions"
        rows = int(lib.silo_gpu_store_scan_rows(store.handle, 0, 0, positions))
        assert 0 < rows < 0.1 * positions  # no row at most positions, one or two where lineages differ or the alignment's ends are ragged
        assert int(lib.silo_gpu_store_scan_runs(store.handle, 0)) > 0
        adaptive = scan_table(lib, store, filt, 0, positions)
        mask = member[lineage].astype(bool)
        assert int(adaptive.sum(axis=1).max()) <= int(mask.sum())
        # the reference's algorithm over roaring-format containers (C port) on three windows of positions
        port_filter = cpu_port.Filter(dense.pack_bits(mask), FULL_N)
        for begin, count in ((0, 32), (positions // 2 // 64 * 64, 64), (positions - 32, 32)):
            port = cpu_port.PortStore(FULL_N, begin, count, "nuc", model=model)
            want, _ = port.mutations_scan(port_filter, n_threads=0, grain=max(1, count // 8))
            port.close()
            assert np.array_equal(adaptive[begin:begin + count], want[:, :5]), (begin, count)
        # and the engine's rows come from these counts: minProportion 0.05 of the filtered total per position
        rows = engine.execute_query(bench.make_query())
        by_mutation = {row["mutation"]: row for row in rows}
        reference = model.reference
        expected = 0
        for position in range(positions):
            total = int(adaptive[position].sum())
            if total == 0:
                continue
            threshold = int(np.ceil(total * 0.05) - 1)
            for symbol in range(5):
                if symbol != reference[position] and adaptive[position][symbol] > threshold:
                    expected += 1
                    row = by_mutation[f"{NUC_CHARS[reference[position]]}{position + 1}{NUC_CHARS[symbol]}"]
                    assert row["count"] == int(adaptive[position][symbol]) and row["proportion"] == adaptive[position][symbol] / total
        assert expected == len(rows) > 100
    finally:
        lib.silo_gpu_free(filt)
    # A second store of the same data that keeps its build-time identity planes (re-encoding switched off): another layout,
    # another kernel instantiation, the whole genome — the same table.  (The module's engine is closed first: both do not
    # fit the device together with their build-time planes.)
    engine.close()
    previous = lib.silo_gpu_tune(4, -1)  # SILO_GPU_TUNE_COMPACT_INDEX < 0: finalize keeps the identity planes
    try:
        plain_engine = bench.build_engine(FULL_N, 0, 1, None, 0)[0]
    finally:
        lib.silo_gpu_tune(4, previous)
    try:
        plain_store = plain_engine.partition_store(0)
        assert lib.silo_gpu_store_scan_planes(plain_store.handle, 0) == 3
        plain_filter = lineage_filter(plain_store.handle)
        identity = scan_table(lib, plain_store, plain_filter, 0, positions)
        lib.silo_gpu_free(plain_filter)
        assert np.array_equal(adaptive, identity)
    finally:
        plain_engine.close()


def test_config3_amino_acid_scan_at_10m_layouts_and_c_port_agree(built):
    """The amino-acid leg of configs[3] at its full size: the 12 genes of 10 M sequences re-encoded at finalize (one-hot rows,
    2 / 3 code planes, identity planes — whatever each position got, 230 M escape keys in 77 slices) against the same genes
    kept on their 5 identity planes (k_scan_sliced<5,22>: another layout, another kernel, no keys), whole genes; windows of S
    and E against the C port of the reference algorithm; the AminoAcidMutations query of the bench on both engines."""
    from silo_amd import alphabet

    lib = binding.load_library()
    engine, _, tree, lineage, _ = bench.build_engine(FULL_N, 0, 1, None, 0, with_genes=True, nuc_positions=64)
    previous = lib.silo_gpu_tune(4, -1)  # SILO_GPU_TUNE_COMPACT_INDEX < 0: finalize keeps the identity planes
    try:
        plain_engine = bench.build_engine(FULL_N, 0, 1, None, 0, with_genes=True, nuc_positions=64)[0]
    finally:
        lib.silo_gpu_tune(4, previous)
    try:
        store, plain_store = engine.partition_store(0), plain_engine.partition_store(0)
        member = tree.subtree(tree.names.index(bench.QUERY_LINEAGE))
        mask = member[lineage].astype(bool)
        filters = []
        for handle in (store.handle, plain_store.handle):
            filt = ctypes.c_void_p()
            binding._check(lib.silo_gpu_bitset_alloc(handle, ctypes.byref(filt)))
            binding._check(lib.silo_gpu_bitset_from_lineages(handle, filt, member.ctypes.data_as(ctypes.c_void_p), len(member), None))
            filters.append(filt)
        genes = bench.load_reference_genomes(True)["genes"]
        port_filter = cpu_port.Filter(dense.pack_bits(mask), FULL_N)
        rows_seen = set()
        for index, gene in enumerate(genes):
            sid = engine.seqstore_id(0, gene["name"], True)
            length = len(gene["sequence"])
            rows = int(lib.silo_gpu_store_scan_rows(store.handle, sid, 0, length))
            assert int(lib.silo_gpu_store_scan_rows(plain_store.handle, sid, 0, length)) == 5 * length
            assert rows < 5 * length or int(lib.silo_gpu_store_scan_escapes(store.handle, sid)) == 0
            rows_seen.add(round(rows / length))
            adaptive = scan_table(lib, store, filters[0], 0, length, sid, 22)
            identity = scan_table(lib, plain_store, filters[1], 0, length, sid, 22)
            assert np.array_equal(adaptive, identity), gene["name"]
            assert int(adaptive.sum(axis=1).max()) <= int(mask.sum())
            if gene["name"] in ("S", "E"):  # the reference's algorithm over roaring-format containers on a window
                reference = np.array([alphabet.AMINO_ACID.char_to_symbol[c] for c in gene["sequence"]], dtype=np.uint8)
                model = synth.make_model(FULL_N, reference, "aa", tree, lineage, seed=synth.DEFAULT_SEED, store_index=index + 1)
                begin, count = (600, 24) if gene["name"] == "S" else (40, 24)
                port = cpu_port.PortStore(FULL_N, begin, count, "aa", model=model)
                want, _ = port.mutations_scan(port_filter, n_threads=0, grain=max(1, count // 8))
                port.close()
                scan_symbols = list(alphabet.AMINO_ACID.valid_mutation_symbols)
                assert np.array_equal(adaptive[begin:begin + count], want[:, scan_symbols]), gene["name"]
        assert len(rows_seen) >= 3  # genes of about 1, 2-3 and 5 plane rows per position: every layout took part
        query = json.dumps({"action": {"type": "AminoAcidMutations", "minProportion": 0.05},
                            "filterExpression": json.loads(bench.make_query())["filterExpression"]})
        rows = engine.execute_query(query)
        assert rows == plain_engine.execute_query(query) and len(rows) > 100
        for filt in filters:
            lib.silo_gpu_free(filt)
    finally:
        engine.close()
        plain_engine.close()


def test_config4_batch_at_6m_rows_matches_the_c_port_on_windows(built):
    """BASELINE.json configs[4] as one of its 8 sequence-id shards holds it — 6.25 M sequences, genome + 12 genes — and as
    bench.py runs it: ONE silo_engine_execute_batch of 100 lineage filters (every fourth AND a nucleotide predicate) x
    (Mutations + AminoAcidMutations), 8 filters per pass over the plane rows and over the escape keys
    (k_scan_sliced<2,2,4,8,rows>, k_scan_escapes_sliced<8>).  Responses of seven of the queries — plain lineage filters of
    different sizes, two with the nucleotide predicate, two amino-acid ones — against the C port of the reference's algorithm
    (oracle/roaring_port.c over roaring-format containers) on windows of positions: counts, thresholds, rows, proportions."""
    from silo_amd import alphabet

    n = 6_250_000
    engine, model, tree, lineage, _ = bench.build_engine(n, 0, 1, None, 0, with_genes=True)
    try:
        reference_text = bench.load_reference_genomes(False)["nucleotideSequences"][0]["sequence"]
        queries = bench.config4_queries(tree, reference_text)
        assert len(queries) == 200
        batched = engine.execute_batch_text(queries)
        assert all(status == 200 for status, _ in batched)
        assert sum(1 for _, body in batched if len(json.loads(body.decode())["queryResult"]) > 0) >= 190
        genes = bench.load_reference_genomes(True)["genes"]
        gene_index = {gene["name"]: k for k, gene in enumerate(genes)}

        def mask_of(expression):
            if expression["type"] == "And":
                return np.logical_and.reduce([mask_of(child) for child in expression["children"]])
            if expression["type"] == "PangoLineage":
                return tree.subtree(tree.names.index(expression["value"]))[lineage].astype(bool)
            column = oracle_synth.symbol_matrix(model, np.arange(n), np.array([expression["position"] - 1]))[:, 0]
            return column == NUC_CHARS.index(expression["symbol"])

        def rows_in_window(document, name, begin, count):
            import re
            picked = []
            for row in document["queryResult"]:
                position = int(re.match(r"^[^0-9]*([0-9]+)", row["mutation"]).group(1)) - 1
                if row["sequenceName"] == name and begin <= position < begin + count:
                    picked.append(row)
            return picked

        def expected_rows(table, reference, valid_symbols, chars, begin, name):
            rows = []
            for offset in range(table.shape[0]):
                total = int(table[offset].sum())
                if total == 0:
                    continue
                must_exceed = int(np.ceil(total * 0.05) - 1)
                for k, symbol in enumerate(valid_symbols):
                    count = int(table[offset][k])
                    if symbol != reference[begin + offset] and count > must_exceed:
                        rows.append({"count": count, "mutation": f"{chars[reference[begin + offset]]}{begin + offset + 1}{chars[symbol]}",
                                     "proportion": count / total, "sequenceName": name})
            return rows

        checked = 0
        for index in (0, 6, 14, 38, 198):  # Mutations queries: lineages of different sizes; 6 and 14 ... 198 carry the nucleotide predicate (k % 4 == 3)
            query = json.loads(queries[index].decode())
            assert query["action"]["type"] == "Mutations"
            mask = mask_of(query["filterExpression"])
            port_filter = cpu_port.Filter(dense.pack_bits(mask), n)
            document = json.loads(batched[index][1].decode())
            for begin, count in ((0, 32), (model.positions // 2 // 64 * 64, 32), (model.positions - 32, 32)):
                port = cpu_port.PortStore(n, begin, count, "nuc", model=model)
                table, _ = port.mutations_scan(port_filter, n_threads=0, grain=max(1, count // 8))
                port.close()
                want = expected_rows(table[:, :5].astype(np.int64), model.reference, [0, 1, 2, 3, 4], NUC_CHARS, begin, "main")
                assert rows_in_window(document, "main", begin, count) == want, (index, begin)
                checked += len(want)
        aa_chars = "-ACDEFGHIKLMNPQRSTVWYBZ*X"
        valid_aa = list(alphabet.AMINO_ACID.valid_mutation_symbols)
        for index, gene_name, begin, count in ((1, "S", 600, 24), (39, "ORF1a", 2000, 24)):
            query = json.loads(queries[index].decode())
            assert query["action"]["type"] == "AminoAcidMutations"
            mask = mask_of(query["filterExpression"])
            gene = genes[gene_index[gene_name]]
            reference = np.array([alphabet.AMINO_ACID.char_to_symbol[c] for c in gene["sequence"]], dtype=np.uint8)
            gene_model = synth.make_model(n, reference, "aa", tree, lineage, seed=synth.DEFAULT_SEED, store_index=gene_index[gene_name] + 1)
            port = cpu_port.PortStore(n, begin, count, "aa", model=gene_model)
            table, _ = port.mutations_scan(cpu_port.Filter(dense.pack_bits(mask), n), n_threads=0, grain=max(1, count // 8))
            port.close()
            want = expected_rows(table[:, valid_aa].astype(np.int64), reference, valid_aa, aa_chars, begin, gene_name)
            document = json.loads(batched[index][1].decode())
            assert rows_in_window(document, gene_name, begin, count) == want, (index, gene_name)
            checked += len(want)
        assert checked > 20  # rows were compared, not empty windows
    finally:
        engine.close()
