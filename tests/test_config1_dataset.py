"""BASELINE.json configs[0]: testBaseData/exampleDataset1000Sequences, Aggregated count with one NucleotideEquals
filter.  No reference test preprocesses this data set (record 0 has a null primary key which the reference's
preprocessor rejects); rows are keyed by file order and all 1000 records are kept.  The expected
counts are the data-derived known answers of SURVEY.md §8c (naive counting over the ndjson)."""
import json
import lzma
import os

import pytest

from oracle import silo_oracle as so
from tests import dataset

KNOWN = {(241, "T"): 740, (3037, "T"): 749, (14408, "T"): 743, (23403, "G"): 748, (28881, "A"): 634, (122, "N"): 1,
         (1, "-"): 988, (29903, "-"): 990, (8782, "T"): 212, (28144, "C"): 212}


def load():
    root = os.path.join(dataset.GOLDEN, "exampleDataset1000Sequences")
    genomes = json.load(open(os.path.join(root, "reference_genomes.json")))
    lineages, sequences = [], []
    with lzma.open(os.path.join(root, "main_aligned.txt.xz"), "rt") as handle:
        for line in handle:
            lineage, _, seq = line.rstrip("\n").partition("\t")
            lineages.append(lineage or None)
            sequences.append(seq or None)
    return genomes, lineages, sequences


def query(position, symbol):
    return {"action": {"type": "Aggregated"}, "filterExpression": {"type": "NucleotideEquals", "position": position, "symbol": symbol}}


def test_naive_counts_are_the_known_answers():
    _, _, sequences = load()
    assert len(sequences) == 1000
    for (position, symbol), want in KNOWN.items():
        assert sum(1 for s in sequences if s is not None and s[position - 1] == symbol) == want


def test_oracle_counts():
    genomes, lineages, sequences = load()
    main = next(g for g in genomes["nucleotideSequences"] if g["name"] == "main")["sequence"]
    db = so.Database({"main": [so.Nucleotide.char_to_symbol(c) for c in main]}, {})
    db.add_partition({"main": sequences}, {}, [l or "" for l in lineages])
    for (position, symbol), want in KNOWN.items():
        assert so.execute_query(db, query(position, symbol)) == [{"count": want}]


@pytest.mark.gpu
def test_engine_counts(built):
    from silo_amd.engine import Engine

    genomes, lineages, sequences = load()
    doc = {"nucleotideSequences": [g for g in genomes["nucleotideSequences"] if g["name"] == "main"], "genes": []}
    with Engine(doc) as engine:
        part = engine.add_partition(len(sequences))
        engine.append_sequences(part, "main", False, 0, sequences)
        engine.set_lineage_column(part, "pango_lineage", lineages)
        engine.finalize()
        for (position, symbol), want in KNOWN.items():
            assert engine.execute_query(query(position, symbol)) == [{"count": want}]
        rows = engine.execute_query({"action": {"type": "Mutations", "minProportion": 0.5}, "filterExpression": {"type": "True"}})
        by_name = {row["mutation"]: row["count"] for row in rows}
        assert by_name["C241T"] == 740 and by_name["C3037T"] == 749 and by_name["A23403G"] == 748
