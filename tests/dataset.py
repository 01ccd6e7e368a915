"""Loads the committed copy of the reference's e2e test data set (tests/golden/exampleDataset)."""
import json
import lzma
import os

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def read_fasta_xz(path):
    out = {}
    key = None
    with lzma.open(path, "rt") as handle:
        for line in handle:
            line = line.rstrip("\n")
            if line.startswith(">"):
                key = line[1:]
            elif key is not None:
                out[key] = line
                key = None
    return out


def load_example_dataset():
    """Rows in metadata-file order (the reference orders rows by partition / date / key through DuckDB,
    SURVEY.md §8c: counts do not depend on the order, id sets are compared oracle-vs-device only)."""
    root = os.path.join(GOLDEN, "exampleDataset")
    genomes = json.load(open(os.path.join(root, "reference_genomes.json")))
    alias = json.load(open(os.path.join(root, "pangolineage_alias.json")))
    with open(os.path.join(root, "small_metadata_set.tsv")) as handle:
        header = handle.readline().rstrip("\n").split("\t")
        rows = [dict(zip(header, line.rstrip("\n").split("\t"))) for line in handle if line.strip()]
    keys = [row["gisaid_epi_isl"] for row in rows]
    nuc, aa = {}, {}
    for entry in genomes["nucleotideSequences"]:
        fasta = read_fasta_xz(os.path.join(root, f"nuc_{entry['name']}.fasta.xz"))
        nuc[entry["name"]] = [fasta.get(key) for key in keys]
    for entry in genomes["genes"]:
        fasta = read_fasta_xz(os.path.join(root, f"gene_{entry['name']}.fasta.xz"))
        aa[entry["name"]] = [fasta.get(key) for key in keys]
    unaligned = {}
    for entry in genomes["nucleotideSequences"]:
        path = os.path.join(root, f"unaligned_{entry['name']}.fasta.xz")
        if os.path.exists(path):
            fasta = read_fasta_xz(path)
            unaligned[entry["name"]] = [fasta.get(key) for key in keys]
    return dict(
        unaligned=unaligned,
        nuc_references={e["name"]: e["sequence"] for e in genomes["nucleotideSequences"]},
        aa_references={e["name"]: e["sequence"] for e in genomes["genes"]},
        nuc=nuc,
        aa=aa,
        lineages=[row["pango_lineage"] for row in rows],
        alias=alias,
        keys=keys,
        rows=rows,
    )


def load_database_config(path=None):
    """The slice of database_config.yaml the engine needs: [(column, type)] in file order with the reference's column
    types (string + generateIndex -> indexed_string, database_config.cpp:158-189), primaryKey, dateToSortBy."""
    import yaml

    path = path or os.path.join(GOLDEN, "exampleDataset", "database_config.yaml")
    schema = yaml.safe_load(open(path))["schema"]
    kinds = {"string": "string", "date": "date", "pango_lineage": "pango_lineage", "int": "int", "float": "float",
             "insertion": "insertion", "aaInsertion": "aa_insertion"}
    metadata = []
    for entry in schema["metadata"]:
        kind = kinds[entry["type"]]
        if kind == "string" and entry.get("generateIndex"):
            kind = "indexed_string"
        metadata.append((entry["name"], kind))
    return dict(metadata=metadata, primary_key=schema["primaryKey"], date_to_sort_by=schema.get("dateToSortBy"))


def load_query_fixtures(kind="queries"):
    root = os.path.join(GOLDEN, kind)
    out = []
    for name in sorted(os.listdir(root)):
        case = json.load(open(os.path.join(root, name)))
        case["file"] = name
        out.append(case)
    return out


# Reference defect pinned by a golden (details.cpp:118-134): in produceSortedTuplesWithLimit the first row past the
# first `limit + offset` rows of a partition is offered to the top-k heap TWICE (once before the loop, once by the
# loop's first iteration), so it can appear twice in the sorted prefix and shifts every later row by one.  Which row
# that is depends on the physical row order DuckDB gave the reference's partitions, which cannot be
# reproduced here; this build returns the intended (duplicate-free) rows.  For this fixture the expected rows are
# the intended result shifted by exactly one position.
DETAILS_DUPLICATE_DEFECT = {"DetailsOrderByLimit.json": 1}


# Goldens without orderByFields whose row order is the physical row order of the reference's partitions (DuckDB-derived,
# SURVEY.md §8c): compared as multisets.
UNORDERED_RESULTS = {"fasta_manySequences.json"}


def check_next_row_case(case, execute):
    """Compares `execute(query)` with the golden; the fixtures in DETAILS_DUPLICATE_DEFECT are compared against the
    intended result shifted by the duplicated row."""
    shift = DETAILS_DUPLICATE_DEFECT.get(case["file"])
    if case["file"] in UNORDERED_RESULTS:
        got, want = execute(case["query"]), case["expectedQueryResult"]
        assert sorted(json.dumps(row, sort_keys=True) for row in got) == sorted(json.dumps(row, sort_keys=True) for row in want)
        return
    if shift is None:
        assert execute(case["query"]) == case["expectedQueryResult"]
        return
    query = json.loads(json.dumps(case["query"]))
    offset, limit = query["action"]["offset"], query["action"]["limit"]
    assert execute(query) != case["expectedQueryResult"]  # if this starts to hold, the exclusion is obsolete
    query["action"]["offset"] = offset - shift
    assert execute(query) == case["expectedQueryResult"]
