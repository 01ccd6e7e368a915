"""Pins the CPU oracle against the reference's own fixtures: the e2e query goldens on
testBaseData/exampleDataset and the operator unit-test vectors."""
import json
import os

import pytest

from oracle import silo_oracle as so
from tests import dataset


def build_oracle_db(data, partition_sizes=None):
    db = so.Database(
        {k: [so.Nucleotide.char_to_symbol(c) for c in v] for k, v in data["nuc_references"].items()},
        {k: [so.AminoAcid.char_to_symbol(c) for c in v] for k, v in data["aa_references"].items()},
        alias_key=data["alias"],
    )
    config = dataset.load_database_config()
    db.set_config(config["metadata"], config["primary_key"], config["date_to_sort_by"])
    n = len(data["keys"])
    bounds = [0]
    for size in partition_sizes or [n]:
        bounds.append(bounds[-1] + size)
    assert bounds[-1] == n
    for lo, hi in zip(bounds[:-1], bounds[1:]):
        partition = db.add_partition(
            {k: v[lo:hi] for k, v in data["nuc"].items()},
            {k: v[lo:hi] for k, v in data["aa"].items()},
            data["lineages"][lo:hi],
        )
        db.add_metadata(partition, data["rows"][lo:hi])
        for name, sequences in data["unaligned"].items():
            partition.unaligned_nuc_sequences[name] = sequences[lo:hi]
    return db


@pytest.fixture(scope="module")
def example_data():
    return dataset.load_example_dataset()


@pytest.fixture(scope="module", params=[None, [37, 1, 62]], ids=["1-partition", "3-partitions"])
def oracle_db(request, example_data):
    return build_oracle_db(example_data, request.param)


def normalise(rows):
    return json.loads(json.dumps(rows))


@pytest.mark.parametrize("case", dataset.load_query_fixtures("queries"), ids=lambda c: c["file"])
def test_e2e_query_goldens(oracle_db, case):
    got = normalise(so.execute_query(oracle_db, case["query"]))
    assert got == case["expectedQueryResult"]


@pytest.mark.parametrize("case", dataset.load_query_fixtures("invalidQueries"), ids=lambda c: c["file"])
def test_e2e_invalid_query_goldens(oracle_db, case):
    with pytest.raises(so.QueryParseException) as info:
        so.execute_query(oracle_db, case["query"])
    assert {"error": "Bad request", "message": str(info.value)} == case["expectedError"]


def as_multiset(rows):
    return sorted(json.dumps(row, sort_keys=True) for row in rows)


@pytest.mark.parametrize("case", dataset.load_query_fixtures("queries_next"), ids=lambda c: c["file"])
def test_e2e_next_row_query_goldens(oracle_db, case):
    """metadata predicates, Aggregated with groupByFields, Details, FastaAligned."""
    dataset.check_next_row_case(case, lambda query: normalise(so.execute_query(oracle_db, query)))


@pytest.mark.parametrize("case", dataset.load_query_fixtures("invalidQueries_next"), ids=lambda c: c["file"])
def test_e2e_next_row_invalid_query_goldens(oracle_db, case):
    with pytest.raises(so.QueryParseException) as info:
        so.execute_query(oracle_db, case["query"])
    assert {"error": "Bad request", "message": str(info.value)} == case["expectedError"]


def test_inline_error_cases(oracle_db):  # endToEndTests/test/query.test.js:80-113
    with pytest.raises(so.QueryParseException, match="Query json must contain filterExpression and action."):
        so.execute_query(oracle_db, {"someJson": "but missing expected properties"})
    with pytest.raises(so.QueryParseException, match="Unknown object filter type 'invalid filter type'"):
        so.execute_query(oracle_db, {"action": {"type": "invalid action"}, "filterExpression": {"type": "invalid filter type"}})


VECTORS = json.load(open(os.path.join(dataset.GOLDEN, "operators", "operator_vectors.json")))


def scans(sets, row_count):
    return [so.IndexScan(so.bits_from_ids(s), row_count) for s in sets]


def clipped(bits, row_count):
    return so.ids_from_bits(bits & ((1 << row_count) - 1))


def clip(ids, row_count):
    return [i for i in ids if i < row_count]


@pytest.mark.parametrize("vec", VECTORS["threshold"], ids=lambda v: v["cite"])
def test_threshold_vectors(vec):
    rc = vec["row_count"]
    for case in vec["cases"]:
        op = so.Threshold(scans(vec["non_negated"], rc), scans(vec["negated"], rc), case["n"], case["exact"], rc)
        assert clipped(op.evaluate(), rc) == clip(case["expected"], rc), case


def test_threshold_and_intersection_invalid():
    for vec in VECTORS["threshold_invalid"]:
        with pytest.raises(so.QueryCompilationException):
            so.Threshold(scans(vec["non_negated"], 0), scans(vec["negated"], 0), vec["n"], vec["exact"], vec["row_count"])
    for vec in VECTORS["intersection_invalid"]:
        with pytest.raises(so.QueryCompilationException):
            so.Intersection(scans(vec["non_negated"], 5), scans(vec["negated"], 5), vec["row_count"])


@pytest.mark.parametrize("vec", VECTORS["intersection"], ids=lambda v: v["cite"])
def test_intersection_vectors(vec):
    rc = vec["row_count"]
    op = so.Intersection(scans(vec["non_negated"], rc), scans(vec["negated"], rc), rc)
    assert so.ids_from_bits(op.evaluate()) == vec["expected"]


@pytest.mark.parametrize("vec", VECTORS["union"], ids=lambda v: v["cite"])
def test_union_vectors(vec):
    rc = vec["row_count"]
    assert so.ids_from_bits(so.Union(scans(vec["children"], rc), rc).evaluate()) == vec["expected"]


@pytest.mark.parametrize("vec", VECTORS["complement"], ids=lambda v: v["cite"])
def test_complement_vectors(vec):
    rc = vec["row_count"]
    op = so.Complement(so.IndexScan(so.bits_from_ids(vec["child"]), rc), rc)
    assert so.ids_from_bits(op.evaluate()) == vec["expected"]


def test_bitmap_selection_vectors():
    for vec in VECTORS["bitmap_selection"]:
        rows = [set(r) for r in vec["rows"]]
        op = so.BitmapSelection(rows, len(rows), so.BitmapSelection.CONTAINS, vec["value"])
        assert so.ids_from_bits(op.evaluate()) == vec["contains"]
        assert so.ids_from_bits(op.negate().evaluate()) == vec["not_contains"]
        assert so.ids_from_bits(op.negate().negate().evaluate()) == vec["contains"]


def test_position_vectors():
    chars = so.Nucleotide.CHARS
    for vec in VECTORS["position"]:
        position = so.Position(so.Nucleotide)
        for char, ids in vec["symbols"].items():
            position.add_values(chars.index(char), ids, 0, vec["sequence_count"])
        if "deleted" in vec:
            assert position.delete_most_numerous(vec["sequence_count"]) == chars.index(vec["deleted"])
            assert position.bitmaps[chars.index(vec["deleted"])] == 0
            with pytest.raises(RuntimeError):
                position.delete_most_numerous(vec["sequence_count"])
        if "flipped" in vec:
            assert position.flip_most_numerous(vec["sequence_count"]) == chars.index(vec["flipped"])
            for char, ids in vec["stored_after_flip"].items():
                assert so.ids_from_bits(position.bitmaps[chars.index(char)]) == ids
            assert position.flip_most_numerous(vec["sequence_count"]) is None


# ---- vectors of the reference's unit tests for the §8(f) row-3 pieces ------------------------------------------------
@pytest.mark.parametrize("vec", VECTORS["range_selection"], ids=lambda v: v["cite"])
def test_range_selection_vectors(vec):
    op = so.RangeSelection([tuple(r) for r in vec["ranges"]], vec["row_count"])
    assert so.ids_from_bits(op.evaluate()) == vec["expected"]
    assert so.ids_from_bits(op.negate().evaluate()) == vec["negated"]
    assert op.negate().type == so.RANGE_SELECTION


@pytest.mark.parametrize("vec", VECTORS["selection"], ids=lambda v: v["cite"])
def test_selection_vectors(vec):
    op = so.Selection([so.Predicate(vec["column"], vec["comparator"], vec["value"])], len(vec["column"]))
    assert op.type == so.SELECTION
    assert so.ids_from_bits(op.evaluate()) == vec["expected"]
    assert so.ids_from_bits(op.negate().evaluate()) == vec["negated"]


def test_insertion_search_vectors():
    vec = VECTORS["insertion_search"]
    db = so.Database({"main": [1] * 4}, {}, default_nucleotide_sequence=vec["default_sequence"])
    db.set_config([("insertions", "insertion")], "key")
    partition = so.DatabasePartition(sequence_count=len(vec["rows"]))
    db.partitions.append(partition)
    db.add_metadata(partition, [{"insertions": value} for value in vec["rows"]])
    for search in vec["searches"]:
        expression = so.InsertionContains(so.Nucleotide, ["insertions"], None, search["position"], search["pattern"])
        assert so.ids_from_bits(expression.compile(db, partition, so.NONE).evaluate()) == search["expected"], search


def test_date_vectors():
    for text, value in VECTORS["dates"]["parse"]:
        assert so.string_to_date(text) == value, text
    for text, printed in VECTORS["dates"]["reprint"]:
        assert so.date_to_string(so.string_to_date(text)) == printed, text


def test_lineage_alias_vectors(example_data):
    vectors = VECTORS["lineage_alias"]
    lookup = so.PangoLineageAliasLookup(vectors["unalias"]["alias_key"])
    for text, expected in vectors["unalias"]["cases"]:
        assert lookup.unalias(text) == expected, text
    lookup = so.PangoLineageAliasLookup(vectors["alias"]["alias_key"])
    for text, expected in vectors["alias"]["cases"]:
        assert lookup.alias(text) == expected, text
    lookup = so.PangoLineageAliasLookup(example_data["alias"])
    for text, expected in vectors["example_file"]:
        assert lookup.unalias(text) == expected, text


@pytest.mark.parametrize("vec", VECTORS["pango_lineage_column"], ids=lambda v: v["cite"])
def test_pango_lineage_column_vectors(vec):
    column = so.PangoLineageColumnPartition(so.PangoLineageAliasLookup({}))
    for value in vec["rows"]:
        column.insert(value)
    for value, including_sublineages, expected in vec["queries"]:
        bits = column.filter_including_sublineages(value) if including_sublineages else column.filter(value)
        assert so.ids_from_bits(bits or 0) == expected, (value, including_sublineages)
        assert (bits is None) == (expected == [])  # std::nullopt for a value that was never indexed


def test_sublineage_relation_vectors():
    for lineage, other, expected in VECTORS["sublineage_relation"]["cases"]:  # isSublineageOf = `other` is one of the parent lineages
        assert (other in so.get_parent_lineages(lineage)) == expected, (lineage, other)


def test_leaf_operator_vectors():
    for vec in VECTORS["leaf_operators"]:
        rc = vec["row_count"]
        op = {"Full": lambda: so.Full(rc), "Empty": lambda: so.Empty(rc),
              "IndexScan": lambda: so.IndexScan(so.bits_from_ids(vec.get("bitmap", [])), rc)}[vec["operator"]]()
        assert so.ids_from_bits(op.evaluate()) == vec["expected"], vec["cite"]


def test_symbol_conversion_vectors():
    vec = VECTORS["symbol_conversion"]
    for key, alphabet in (("nucleotide", so.Nucleotide), ("amino_acid", so.AminoAcid)):
        gap = alphabet.char_to_symbol("-")
        for char in vec[key]["gap_characters"]:
            assert alphabet.char_to_symbol(char) == gap
        for char in vec[key]["legal"]:
            assert alphabet.char_to_symbol(char) is not None and alphabet.char_to_symbol(char) != gap
        for char in vec[key]["illegal"]:
            assert alphabet.char_to_symbol(char) is None
