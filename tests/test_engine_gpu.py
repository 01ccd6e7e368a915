"""End-to-end parity of the C++ QueryEngine mirror + HIP kernels: the reference's own e2e goldens on
testBaseData/exampleDataset, and oracle-vs-device on randomised queries."""
import json
import random

import pytest

pytestmark = pytest.mark.gpu

from oracle import silo_oracle as so  # noqa: E402
from tests import dataset  # noqa: E402
from tests.test_oracle_golden import build_oracle_db  # noqa: E402


def build_engine(data, partition_sizes=None):
    from silo_amd.engine import Engine

    genomes = json.load(open(dataset.GOLDEN + "/exampleDataset/reference_genomes.json"))
    engine = Engine(genomes, data["alias"])
    config = dataset.load_database_config()
    engine.set_schema(config["primary_key"], config["date_to_sort_by"])
    column_types = {"aa_insertion": "aaInsertion"}
    n = len(data["keys"])
    bounds = [0]
    for size in partition_sizes or [n]:
        bounds.append(bounds[-1] + size)
    for lo, hi in zip(bounds[:-1], bounds[1:]):
        part = engine.add_partition(hi - lo)
        for name, genomes_list in data["nuc"].items():
            engine.append_sequences(part, name, False, 0, genomes_list[lo:hi])
        for name, genomes_list in data["aa"].items():
            engine.append_sequences(part, name, True, 0, genomes_list[lo:hi])
        for name, sequences in data["unaligned"].items():
            engine.append_unaligned_sequences(part, name, sequences[lo:hi])
        for column, kind in config["metadata"]:  # the lineage column also feeds the PangoLineage filter
            engine.append_metadata(part, column, column_types.get(kind, kind), [row.get(column) or "" for row in data["rows"][lo:hi]])
    engine.finalize()
    return engine


@pytest.fixture(scope="module")
def example_data():
    return dataset.load_example_dataset()


@pytest.fixture(scope="module", params=[None, [37, 1, 62]], ids=["1-partition", "3-partitions"])
def engines(request, built, example_data):
    engine = build_engine(example_data, request.param)
    oracle_db = build_oracle_db(example_data, request.param)
    yield engine, oracle_db
    engine.close()


@pytest.mark.parametrize("case", dataset.load_query_fixtures("queries"), ids=lambda c: c["file"])
def test_reference_e2e_goldens(engines, case):
    engine, _ = engines
    status, document = engine.execute_raw(case["query"])
    assert status == 200, document
    assert document == {"queryResult": case["expectedQueryResult"]}


@pytest.mark.parametrize("case", dataset.load_query_fixtures("invalidQueries"), ids=lambda c: c["file"])
def test_reference_e2e_invalid_goldens(engines, case):
    engine, _ = engines
    status, document = engine.execute_raw(case["query"])
    assert status == 400
    assert document == case["expectedError"]


def test_inline_error_cases(engines):  # endToEndTests/test/query.test.js:64-113
    engine, _ = engines
    assert engine.data_version().isdigit() and int(engine.data_version()) > 1_600_000_000  # headerToHaveDataVersion, common.js
    status, document = engine.execute_raw({"someJson": "but missing expected properties"})
    assert (status, document) == (400, {"error": "Bad request", "message": "Query json must contain filterExpression and action."})
    status, document = engine.execute_raw({"action": {"type": "invalid action"}, "filterExpression": {"type": "invalid filter type"}})
    assert (status, document) == (400, {"error": "Bad request", "message": "Unknown object filter type 'invalid filter type'"})
    status, document = engine.execute_raw("{ not a valid json")
    assert status == 400 and document["error"] == "Bad request"
    assert document["message"].startswith("The query was not a valid JSON: ")
    # std::out_of_range from .at() is a 500 in the reference too (has_mutation.cpp:46-47)
    status, document = engine.execute_raw({"action": {"type": "Aggregated"}, "filterExpression": {"type": "HasNucleotideMutation", "position": 0}})
    assert status == 500 and document["error"] == "Internal Server Error"


NUC = "ACGT-NRYKM"
AA = "ACDEFGHIKLMNPQRSTVWY-*XBZ"


def random_leaf(rng):
    kind = rng.choice(["nuc", "nuc", "nuc", "aa", "hasnuc", "hasaa", "lineage", "true", "false"])
    if kind == "nuc":
        store = rng.choice(["main", "main", None, "testSecondSequence"])
        if store == "testSecondSequence":
            pos = rng.randint(1, 4)
        else:
            pos = rng.choice([1, 2, 241, 3037, 14408, 23403, 27542, 28881, 29903])
        expr = {"type": "NucleotideEquals", "position": pos, "symbol": rng.choice(NUC + ".")}
        if store is not None:
            expr["sequenceName"] = store
        return expr
    if kind == "aa":
        gene, length = rng.choice([("S", 1274), ("E", 76), ("N", 420), ("ORF1a", 4401)])
        pos = rng.choice([1, 2, 19, 501, 614, 681, length])
        return {"type": "AminoAcidEquals", "sequenceName": gene, "position": min(pos, length), "symbol": rng.choice(AA + ".")}
    if kind == "hasnuc":
        return {"type": "HasNucleotideMutation", "position": rng.choice([1, 241, 3037, 23403, 28881, 29903, 210, 4184])}
    if kind == "hasaa":
        gene, length = rng.choice([("S", 1274), ("E", 76), ("N", 420)])
        return {"type": "HasAminoAcidMutation", "sequenceName": gene, "position": rng.choice([1, 19, 69, 76, length])}
    if kind == "lineage":
        return {"type": "PangoLineage", "column": "pango_lineage",
                "value": rng.choice(["B.1.1.7", "b.1", "Q", "B.1.617.2", "XA.1", "AY.4", "nonsense", "B.1.1"]),
                "includeSublineages": rng.random() < 0.6}
    return {"type": "True"} if kind == "true" else {"type": "False"}


def random_expression(rng, depth):
    if depth == 0 or rng.random() < 0.25:
        return random_leaf(rng)
    kind = rng.choice(["And", "Or", "Not", "N-Of", "N-Of", "Maybe", "Exact"])
    if kind in ("And", "Or"):
        return {"type": kind, "children": [random_expression(rng, depth - 1) for _ in range(rng.randint(0, 4))]}
    if kind == "N-Of":
        k = rng.randint(1, 5)
        return {"type": "N-Of", "children": [random_expression(rng, depth - 1) for _ in range(k)],
                "numberOfMatchers": rng.randint(0, k + 1), "matchExactly": rng.random() < 0.5}
    return {"type": kind, "child": random_expression(rng, depth - 1)}


def test_random_filters_match_oracle(engines):
    engine, oracle_db = engines
    rng = random.Random(20250117)
    for trial in range(150):
        expression = random_expression(rng, 3)
        query = {"action": {"type": "Aggregated"}, "filterExpression": expression}
        want = so.execute_query(oracle_db, query)
        got = engine.execute_query(query)
        assert got == want, json.dumps(query)


def test_random_mutations_match_oracle(engines):
    engine, oracle_db = engines
    rng = random.Random(7)
    for trial in range(12):
        expression = random_expression(rng, 2)
        if rng.random() < 0.5:
            action = {"type": "Mutations", "minProportion": rng.choice([0, 0.05, 0.5, 1]),
                      "sequenceName": rng.choice(["main", "testSecondSequence", ["main", "testSecondSequence"]])}
        else:
            action = {"type": "AminoAcidMutations", "minProportion": rng.choice([0, 0.02, 0.3])}
            if rng.random() < 0.6:
                action["sequenceName"] = rng.choice(["S", ["E", "N"], "ORF1a"])
        if rng.random() < 0.5:
            action["orderByFields"] = [{"field": "count", "order": "descending"}, "mutation"]
            action["limit"] = rng.randint(1, 50)
            action["offset"] = rng.randint(0, 5)
        query = {"action": action, "filterExpression": expression}
        want = json.loads(json.dumps(so.execute_query(oracle_db, query)))
        got = engine.execute_query(query)
        assert got == want, json.dumps(query)


def test_concurrent_clients_get_sequential_results(engines):
    """executeQuery is re-entrant on a shared engine (the reference serves one thread per request under a
    shared lock, database_mutex.cpp:20-23); here every thread also has its own HIP stream."""
    import threading

    engine, _ = engines
    rng = random.Random(99)
    queries = []
    for _ in range(24):
        expression = random_expression(rng, 3)
        action = rng.choice([
            {"type": "Aggregated"},
            {"type": "Mutations", "minProportion": 0.05},
            {"type": "AminoAcidMutations", "minProportion": 0.1, "sequenceName": "S"},
        ])
        queries.append({"action": action, "filterExpression": expression})
    expected = [engine.execute_raw(q) for q in queries]
    results = {}

    def client(index):
        mine = []
        for round_ in range(3):
            for k in range(index, len(queries), 4):
                mine.append((k, engine.execute_raw(queries[k])))
        results[index] = mine

    threads = [threading.Thread(target=client, args=(i,)) for i in range(4)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert len(results) == 4
    for mine in results.values():
        for k, got in mine:
            assert got == expected[k], json.dumps(queries[k])


def test_native_client_threads_answer_like_one_call(engines):
    """silo_engine_run_clients (the load generator behind bench.py's one-by-one client figures): every client's answers are
    those of a single silo_engine_execute_query; a query that is not answered with 200 ends the run with an error."""
    engine, _ = engines
    rng = random.Random(11)
    for query in ({"action": {"type": "Aggregated"}, "filterExpression": random_expression(rng, 3)},
                  {"action": {"type": "Mutations", "minProportion": 0.05}, "filterExpression": {"type": "True"}}):
        status, body = engine.execute_text(json.dumps(query))
        assert status == 200
        for clients in (1, 4):
            per_second, last = engine.run_clients(json.dumps(query), clients, 0.2)
            assert per_second > 0 and last == body
    with pytest.raises(Exception):
        engine.run_clients(json.dumps({"action": {"type": "Aggregated"}, "filterExpression": {"type": "NoSuchFilter"}}), 2, 0.1)


def test_first_count_queries_of_new_threads_on_a_busy_device(engines):
    """A thread's count slot (k_filter_eval hands its total to the host through it) is created by the thread's first filter ->
    count query.  Its counters are zeroed by a fill on the null stream, which the thread's own non-blocking stream does not
    wait for: with the device busy the fill used to land inside the first launch now and then ("the kernel finished without
    delivering its total", tools/soak.py).  Fresh threads, their first queries while other threads keep the device busy."""
    import threading

    engine, _ = engines
    rng = random.Random(7)
    counts = [{"action": {"type": "Aggregated"}, "filterExpression": random_expression(rng, 3)} for _ in range(6)]
    scans = [{"action": {"type": "Mutations", "minProportion": 0.0}, "filterExpression": {"type": "True"}},
             {"action": {"type": "AminoAcidMutations", "minProportion": 0.0}, "filterExpression": {"type": "Not", "child": {"type": "False"}}}]
    expected = [engine.execute_raw(q) for q in counts]
    stop = threading.Event()
    failures = []

    def load():
        while not stop.is_set():
            for q in scans:
                engine.execute_raw(q)

    def newcomer(index):
        for k in range(len(counts)):
            got = engine.execute_raw(counts[(k + index) % len(counts)])
            if got != expected[(k + index) % len(counts)]:
                failures.append((index, k, got))

    background = [threading.Thread(target=load) for _ in range(3)]
    for t in background:
        t.start()
    try:
        for wave in range(6):  # 48 threads, each with a count slot of its own
            fresh = [threading.Thread(target=newcomer, args=(wave * 8 + i,)) for i in range(8)]
            for t in fresh:
                t.start()
            for t in fresh:
                t.join()
    finally:
        stop.set()
        for t in background:
            t.join()
    assert not failures, failures[:3]


def test_full_filter_uses_cached_totals_and_matches_scan(engines):
    engine, oracle_db = engines
    query = {"action": {"type": "Mutations", "minProportion": 0.0, "sequenceName": "main"}, "filterExpression": {"type": "True"}}
    first = engine.execute_query(query)   # computes the totals
    second = engine.execute_query(query)  # adds the cached totals
    want = json.loads(json.dumps(so.execute_query(oracle_db, query)))
    assert first == want and second == want
    # a filter that happens to select every row takes the same path
    everything = {"action": query["action"], "filterExpression": {"type": "Or", "children": [
        {"type": "NucleotideEquals", "position": 5, "symbol": "A"}, {"type": "Not", "child": {"type": "NucleotideEquals", "position": 5, "symbol": "A"}}]}}
    assert engine.execute_query(everything) == want


def test_batch_equals_individual_queries(engines):
    """silo_engine_execute_batch: queries whose scans share passes over the planes (K1c) give exactly the
    documents and statuses of one-by-one execution, in order; a failing query does not disturb the others."""
    engine, oracle_db = engines
    rng = random.Random(4242)
    queries = []
    for _ in range(30):
        expression = random_expression(rng, 2)
        action = rng.choice([
            {"type": "Aggregated"},
            {"type": "Mutations", "minProportion": 0.05},
            {"type": "Mutations", "minProportion": 0.0, "sequenceName": ["main", "testSecondSequence"],
             "orderByFields": [{"field": "count", "order": "descending"}, "mutation"], "limit": 20},
            {"type": "AminoAcidMutations", "minProportion": 0.1, "sequenceName": "S"},
            {"type": "AminoAcidMutations", "minProportion": 0.02},
        ])
        queries.append({"action": action, "filterExpression": expression})
    queries.insert(3, {"action": {"type": "Mutations", "minProportion": 2}, "filterExpression": {"type": "True"}})  # 400
    queries.insert(7, "{ not json")  # 400
    queries.insert(11, {"action": {"type": "Mutations", "sequenceName": "nope"}, "filterExpression": {"type": "True"}})  # 400
    queries.insert(13, {"action": {"type": "Aggregated"}, "filterExpression": {"type": "HasNucleotideMutation", "position": 0}})  # 500
    queries.insert(17, {"action": {"type": "Mutations", "minProportion": 0.0}, "filterExpression": {"type": "True"}})  # cached totals
    expected = [engine.execute_raw(q) for q in queries]
    assert sorted({status for status, _ in expected}) == [200, 400, 500]
    assert engine.execute_batch(queries) == expected
    assert engine.execute_batch([]) == []
    assert engine.execute_batch(queries[:1]) == expected[:1]
    # and against the oracle for the successful ones
    for query, (status, document) in zip(queries, engine.execute_batch(queries)):
        if status == 200:
            assert document["queryResult"] == json.loads(json.dumps(so.execute_query(oracle_db, query))), json.dumps(query)


def test_device_row_selection_equals_host_selection(engines):
    """K4 (k_mutations_select) picks the result rows on the device; with the list capacity at 0 the host selects
    from the whole table, with a tiny capacity the overflow fallback runs: all three give the same documents."""
    engine, oracle_db = engines
    rng = random.Random(31)
    queries = []
    for min_proportion in (0, 0.0001, 0.05, 0.5, 1):
        for action_type, names in (("Mutations", ["main", "testSecondSequence"]), ("AminoAcidMutations", [])):
            action = {"type": action_type, "minProportion": min_proportion}
            if names:
                action["sequenceName"] = names
            queries.append({"action": action, "filterExpression": random_expression(rng, 2)})
            queries.append({"action": action, "filterExpression": {"type": "True"}})
    try:
        documents = {}
        for capacity in (4096, 0, 3, 1 << 20):
            engine.set_option("mutation_row_capacity", capacity)
            documents[capacity] = [engine.execute_raw(q) for q in queries]
            assert engine.execute_batch(queries) == documents[capacity]
    finally:
        engine.set_option("mutation_row_capacity", 4096)
    assert documents[0] == documents[4096] == documents[3] == documents[1 << 20]
    for query, (status, document) in zip(queries, documents[4096]):
        assert status == 200
        assert document["queryResult"] == json.loads(json.dumps(so.execute_query(oracle_db, query))), json.dumps(query)
    with pytest.raises(Exception):
        engine.set_option("no_such_option", 1)


# ---- metadata predicates, Aggregated with groupByFields, Details, FastaAligned -------------
@pytest.mark.parametrize("case", dataset.load_query_fixtures("queries_next"), ids=lambda c: c["file"])
def test_reference_e2e_next_row_goldens(engines, case):
    engine, _ = engines

    def execute(query):
        status, document = engine.execute_raw(query)
        assert status == 200, document
        return document["queryResult"]

    dataset.check_next_row_case(case, execute)


@pytest.mark.parametrize("case", dataset.load_query_fixtures("invalidQueries_next"), ids=lambda c: c["file"])
def test_reference_e2e_next_row_invalid_goldens(engines, case):
    engine, _ = engines
    status, document = engine.execute_raw(case["query"])
    assert status == 400
    assert document == case["expectedError"]


def random_metadata_leaf(rng):
    kind = rng.choice(["string", "indexed", "int_eq", "int_between", "float_eq", "float_between", "date", "unsorted_date", "insertion"])
    if kind == "insertion":
        if rng.random() < 0.5:
            expr = {"type": "InsertionContains", "position": rng.choice([25701, 22339, 22204, 5959, 1]),
                    "value": rng.choice(["CCC", "CC.*", ".*C.*G.*", "TAT", ".*", "G.*T", "AAAA"])}
            if rng.random() < 0.5:
                expr["column"] = "nucleotideInsertions"
            if rng.random() < 0.3:
                expr["sequenceName"] = rng.choice(["main", "testSecondSequence"])
            return expr
        return {"type": "AminoAcidInsertionContains", "position": rng.choice([214, 210, 247, 143]), "sequenceName": rng.choice(["S", "S", "N"]),
                "value": rng.choice(["EPE", "E.*E", ".*", "IV", "T", "SGE.*"]), "column": "aminoAcidInsertions"}
    if kind == "string":
        return {"type": "StringEquals", "column": "gisaid_epi_isl", "value": rng.choice(["EPI_ISL_1749899", "EPI_ISL_1408408", "nope", None])}
    if kind == "indexed":
        column, values = rng.choice([("country", ["Switzerland", "Germany", None, ""]), ("division", ["Bern", "Aargau", "Zürich", None, "x"]),
                                     ("region", ["Europe", "Asia"])])
        return {"type": "StringEquals", "column": column, "value": rng.choice(values)}
    if kind == "int_eq":
        return {"type": "IntEquals", "column": "age", "value": rng.choice([4, 50, 54, 0, None])}
    if kind == "int_between":
        return {"type": "IntBetween", "column": "age", "from": rng.choice([None, 0, 30, 55]), "to": rng.choice([None, 40, 60, 10])}
    if kind == "float_eq":
        return {"type": "FloatEquals", "column": "qc_value", "value": rng.choice([0.9, 0.98, 0.5, None])}
    if kind == "float_between":
        return {"type": "FloatBetween", "column": "qc_value", "from": rng.choice([None, 0.9, 0.95]), "to": rng.choice([None, 0.93, 0.99])}
    column = "date" if kind == "date" else "unsorted_date"
    return {"type": "DateBetween", "column": column, "from": rng.choice([None, "2021-01-01", "2020-11-24", "bogus"]),
            "to": rng.choice([None, "2021-03-18", "2021-06-01", "2020-12-31"])}


def random_mixed_expression(rng, depth):
    if depth == 0 or rng.random() < 0.3:
        return random_metadata_leaf(rng) if rng.random() < 0.6 else random_leaf(rng)
    kind = rng.choice(["And", "And", "Or", "Not", "N-Of", "Maybe"])
    if kind in ("And", "Or"):
        return {"type": kind, "children": [random_mixed_expression(rng, depth - 1) for _ in range(rng.randint(0, 4))]}
    if kind == "N-Of":
        k = rng.randint(1, 4)
        return {"type": "N-Of", "children": [random_mixed_expression(rng, depth - 1) for _ in range(k)],
                "numberOfMatchers": rng.randint(0, k), "matchExactly": rng.random() < 0.5}
    return {"type": kind, "child": random_mixed_expression(rng, depth - 1)}


def as_multiset(rows):
    return sorted(json.dumps(row, sort_keys=True) for row in rows)


def test_random_metadata_filters_match_oracle(engines):
    """Metadata predicates mixed with sequence predicates under And / Or / Not / N-Of (the Selection operator, its
    comparator-negating Not and its merge into And) against the oracle."""
    engine, oracle_db = engines
    rng = random.Random(606)
    for trial in range(200):
        query = {"action": {"type": "Aggregated"}, "filterExpression": random_mixed_expression(rng, 3)}
        assert engine.execute_query(query) == so.execute_query(oracle_db, query), json.dumps(query)


def test_random_group_by_and_details_match_oracle(engines):
    engine, oracle_db = engines
    rng = random.Random(707)
    columns = ["gisaid_epi_isl", "date", "unsorted_date", "region", "country", "pango_lineage", "division", "age", "qc_value",
               "nucleotideInsertions", "aminoAcidInsertions"]
    for trial in range(40):
        expression = random_mixed_expression(rng, 2)
        group_by = rng.sample(columns, rng.randint(1, 3))
        query = {"action": {"type": "Aggregated", "groupByFields": group_by}, "filterExpression": expression}
        want = json.loads(json.dumps(so.execute_query(oracle_db, query)))
        got = engine.execute_query(query)
        assert as_multiset(got) == as_multiset(want), json.dumps(query)  # row order is unspecified without orderByFields
        # Details ordered by the primary key (unique): exact row order
        fields = rng.sample(columns, rng.randint(1, 4))
        action = {"type": "Details", "fields": fields + ["gisaid_epi_isl"],
                  "orderByFields": [{"field": rng.choice(fields), "order": rng.choice(["ascending", "descending"])}, "gisaid_epi_isl"]}
        if rng.random() < 0.5:
            action["limit"] = rng.randint(0, 30)
        if rng.random() < 0.5:
            action["offset"] = rng.randint(0, 10)
        query = {"action": action, "filterExpression": expression}
        want = json.loads(json.dumps(so.execute_query(oracle_db, query)))
        assert engine.execute_query(query) == want, json.dumps(query)


def test_group_by_with_a_huge_tuple_space_matches_oracle(engines):
    """Five fields whose dictionary sizes multiply far past 2^24: the group-by goes through the HBM hash table (K6b)."""
    engine, oracle_db = engines
    fields = ["gisaid_epi_isl", "date", "age", "division", "qc_value", "pango_lineage"]
    for expression in ({"type": "True"}, {"type": "IntBetween", "column": "age", "from": 40, "to": None},
                       {"type": "StringEquals", "column": "country", "value": "nowhere"}):
        query = {"action": {"type": "Aggregated", "groupByFields": fields, "orderByFields": ["gisaid_epi_isl"]}, "filterExpression": expression}
        want = json.loads(json.dumps(so.execute_query(oracle_db, query)))
        assert engine.execute_query(query) == want


def test_random_insertions_actions_match_oracle(engines):
    engine, oracle_db = engines
    rng = random.Random(808)
    for trial in range(30):
        if rng.random() < 0.5:
            action = {"type": "Insertions"}
            if rng.random() < 0.5:
                action["column"] = rng.choice(["nucleotideInsertions", ["nucleotideInsertions"]])
            if rng.random() < 0.4:
                action["sequenceName"] = rng.choice(["main", ["main", "testSecondSequence"]])
        else:
            action = {"type": "AminoAcidInsertions"}
            if rng.random() < 0.5:
                action["column"] = "aminoAcidInsertions"
            if rng.random() < 0.5:
                action["sequenceName"] = rng.choice(["S", ["S", "N"], "ORF1a"])
        query = {"action": action, "filterExpression": random_mixed_expression(rng, 2)}
        want = json.loads(json.dumps(so.execute_query(oracle_db, query)))
        assert as_multiset(engine.execute_query(query)) == as_multiset(want), json.dumps(query)
        ordered = dict(action, orderByFields=["sequenceName", "position", "insertions"])
        query = {"action": ordered, "filterExpression": query["filterExpression"]}
        assert engine.execute_query(query) == json.loads(json.dumps(so.execute_query(oracle_db, query))), json.dumps(query)


def test_fasta_aligned_matches_oracle(engines):
    engine, oracle_db = engines
    query = {"action": {"type": "FastaAligned", "sequenceName": ["main", "S", "testSecondSequence"], "orderByFields": ["gisaid_epi_isl"]},
             "filterExpression": {"type": "IntBetween", "column": "age", "from": 50, "to": 52}}
    want = json.loads(json.dumps(so.execute_query(oracle_db, query)))
    got = engine.execute_query(query)
    assert len(got) > 0 and got == want


def test_insertion_search_unit_test_vectors(built):
    """insertion_column.test.cpp:34-62 through InsertionContains on the device (K8), rows read back with Details."""
    import os

    from silo_amd.engine import Engine

    vec = json.load(open(os.path.join(dataset.GOLDEN, "operators", "operator_vectors.json")))["insertion_search"]
    genomes = {"nucleotideSequences": [{"name": "main", "sequence": "ACGT"}], "genes": []}
    with Engine(genomes) as engine:
        engine.set_schema("key")
        n = len(vec["rows"])
        part = engine.add_partition(n)
        engine.append_sequences(part, "main", False, 0, [None] * n)
        engine.append_metadata(part, "key", "string", [str(i) for i in range(n)])
        engine.append_metadata(part, "insertions", "insertion", vec["rows"])
        engine.finalize()
        for search in vec["searches"]:
            rows = engine.execute_query({"action": {"type": "Details", "fields": ["key"], "orderByFields": ["key"]},
                                         "filterExpression": {"type": "InsertionContains", "column": "insertions",
                                                              "position": search["position"], "value": search["pattern"]}})
            assert [int(row["key"]) for row in rows] == search["expected"], search
        assert engine.execute_query({"action": {"type": "Details", "fields": ["insertions"], "orderByFields": ["insertions"], "limit": 2},
                                     "filterExpression": {"type": "True"}}) == [{"insertions": "25701:ACCA"}, {"insertions": "25701:ACCA"}]


def _keyed_engine(n, extra_columns, lineages=None, sequences=None):
    """A tiny database: `n` rows, a string primary key "0".."n-1", the given metadata columns."""
    from silo_amd.engine import Engine

    engine = Engine({"nucleotideSequences": [{"name": "main", "sequence": "ACGT"}], "genes": []})
    engine.set_schema("key")
    part = engine.add_partition(n)
    engine.append_sequences(part, "main", False, 0, sequences or [None] * n)
    engine.append_metadata(part, "key", "string", [str(i) for i in range(n)])
    for name, kind, values in extra_columns:
        engine.append_metadata(part, name, kind, values)
    if lineages is not None:
        engine.set_lineage_column(part, "pango_lineage", lineages)
    engine.finalize()
    return engine


def _selected_rows(engine, expression):
    rows = engine.execute_query({"action": {"type": "Details", "fields": ["key"]}, "filterExpression": expression})
    return sorted(int(row["key"]) for row in rows)


def test_column_unit_test_vectors(built):
    """pango_lineage_column.test.cpp, indexed_string_column.test.cpp, pango_lineage.test.cpp (isSublineageOf),
    nucleotide_symbols.test.cpp ('.' is a gap) as row sets of filters evaluated on the device."""
    import os

    vectors = json.load(open(os.path.join(dataset.GOLDEN, "operators", "operator_vectors.json")))
    for vec in vectors["pango_lineage_column"]:
        with _keyed_engine(len(vec["rows"]), [], lineages=vec["rows"]) as engine:
            for value, including_sublineages, expected in vec["queries"]:
                got = _selected_rows(engine, {"type": "PangoLineage", "column": "pango_lineage", "value": value,
                                              "includeSublineages": including_sublineages})
                assert got == expected, (vec["cite"], value, including_sublineages)
    for vec in vectors["indexed_string_column"]:
        for kind in ("indexed_string", "string"):
            with _keyed_engine(len(vec["rows"]), [("name", kind, vec["rows"])]) as engine:
                for value, expected in vec["queries"]:
                    assert _selected_rows(engine, {"type": "StringEquals", "column": "name", "value": value}) == expected, (vec["cite"], kind, value)
    cases = vectors["sublineage_relation"]["cases"]
    with _keyed_engine(len(cases), [], lineages=[lineage for lineage, _, _ in cases]) as engine:
        for row, (lineage, other, expected) in enumerate(cases):
            got = _selected_rows(engine, {"type": "PangoLineage", "column": "pango_lineage", "value": other, "includeSublineages": True})
            assert (row in got) == expected, (lineage, other)
    conversion = vectors["symbol_conversion"]["nucleotide"]
    with _keyed_engine(3, [], sequences=["A.GT", "A-GT", "ACGT"]) as engine:  # both gap characters are the same symbol
        assert conversion["gap_characters"] == [".", "-"]
        assert _selected_rows(engine, {"type": "NucleotideEquals", "position": 2, "symbol": "-"}) == [0, 1]
    for char in conversion["illegal"]:
        with pytest.raises(Exception, match="[Ii]llegal|invalid"):
            _keyed_engine(1, [], sequences=["AC" + char + "T"])
    for vec in vectors["leaf_operators"]:
        if vec["operator"] != "IndexScan":
            with _keyed_engine(vec["row_count"], []) as engine:
                expression = {"type": "True"} if vec["operator"] == "Full" else {"type": "False"}
                assert _selected_rows(engine, expression) == vec["expected"], vec["cite"]


def test_compat_remove_quirk_is_a_named_switch(engines):
    """SILO_COMPAT_REMOVE_QUIRK: with std::remove and no erase (has_mutation.cpp:58-65) the reference keeps
    T in the list at a reference-T position, so HasNucleotideMutation there also matches the rows that carry T.  Default
    on = the reference's behaviour (the oracle restates it); switched off, exactly those rows drop out.  No reference
    fixture pins the reference-T case (secondSequenceHasMutation.json sits on a reference-C position): oracle <-> device."""
    engine, oracle_db = engines
    genomes = json.load(open(dataset.GOLDEN + "/exampleDataset/reference_genomes.json"))
    reference = next(g["sequence"] for g in genomes["nucleotideSequences"] if g["name"] == "main")
    checked = 0
    for position in (241, 3037, 23403, 28881, 210, 26767, 27638, 22917):
        has_mutation = {"action": {"type": "Aggregated"}, "filterExpression": {"type": "HasNucleotideMutation", "position": position}}
        carries = lambda symbol: engine.execute_query(  # noqa: E731
            {"action": {"type": "Aggregated"}, "filterExpression": {"type": "NucleotideEquals", "position": position, "symbol": symbol}})[0]["count"]
        with_quirk = engine.execute_query(has_mutation)
        assert with_quirk == so.execute_query(oracle_db, has_mutation)
        try:
            engine.set_option("compat_remove_quirk", 0)
            without_quirk = engine.execute_query(has_mutation)[0]["count"]
        finally:
            engine.set_option("compat_remove_quirk", 1)
        others = sum(carries(s) for s in "ACGT" if s != reference[position - 1])
        assert without_quirk == others
        if reference[position - 1] == "T":
            assert with_quirk[0]["count"] == others + carries("T")  # the reference symbol itself counts as a "mutation"
            checked += 1
        else:
            assert with_quirk[0]["count"] == others
    assert checked >= 2, "no reference-T position among the probed ones"
