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
    return dict(
        nuc_references={e["name"]: e["sequence"] for e in genomes["nucleotideSequences"]},
        aa_references={e["name"]: e["sequence"] for e in genomes["genes"]},
        nuc=nuc,
        aa=aa,
        lineages=[row["pango_lineage"] for row in rows],
        alias=alias,
        keys=keys,
        rows=rows,
    )


def load_query_fixtures(kind="queries"):
    root = os.path.join(GOLDEN, kind)
    out = []
    for name in sorted(os.listdir(root)):
        case = json.load(open(os.path.join(root, name)))
        case["file"] = name
        out.append(case)
    return out
