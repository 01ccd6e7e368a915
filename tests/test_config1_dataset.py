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


def layout_statistics(sequences):
    """Per position of the real alignment: rows whose valid symbol (- A C G T) is none of the position's three most frequent
    — the rows the 2-plane layout of the adaptive code planes has to list as escape keys (csrc/silo_gpu.hip, chooseLayouts)."""
    import numpy as np

    rows = [s for s in sequences if s is not None]
    matrix = np.frombuffer("".join(rows).encode(), dtype=np.uint8).reshape(len(rows), -1)
    counts = np.stack([(matrix == ord(c)).sum(axis=0) for c in "-ACGT"], axis=1)  # [P][5]
    ordered = np.sort(counts, axis=1)[:, ::-1]
    escapes = ordered[:, 3:].sum(axis=1)
    return len(rows), matrix.shape[1], escapes


def one_hot_statistics(sequences):
    """Per position: valid-symbol rows beside the most frequent symbol / beside the two most frequent — what a position
    stored as ONE one-hot row / as two lists as escape keys."""
    import numpy as np

    rows = [s for s in sequences if s is not None]
    matrix = np.frombuffer("".join(rows).encode(), dtype=np.uint8).reshape(len(rows), -1)
    counts = np.stack([(matrix == ord(c)).sum(axis=0) for c in "-ACGT"], axis=1)
    ordered = np.sort(counts, axis=1)[:, ::-1]
    return len(rows), matrix.shape[1], ordered[:, 1:].sum(axis=1), ordered[:, 2:].sum(axis=1)


def test_real_alignment_qualifies_for_the_two_plane_layout():
    """the 2-plane layout rests on 'three symbols cover a position'.  On the 1 000 real SARS-CoV-2
    sequences of exampleDataset1000Sequences: the escape keys are far below the 1/512 of the cells the round-1 index
    budgeted, and all but a handful of positions pass the per-position test of the cost model (escapes <= N / 320)."""
    _, _, sequences = load()
    n, positions, escapes = layout_statistics(sequences)
    fraction = escapes.sum() / (n * positions)
    qualifying = (escapes <= n / 320).mean()
    print(f"escape cells: {int(escapes.sum())} of {n * positions} = {fraction:.2e}; positions with escapes <= N/320: {qualifying:.4f}; "
          f"worst position: {int(escapes.max())} rows")
    assert fraction < 1 / 512 / 20
    assert qualifying > 0.995


def test_real_alignment_is_one_symbol_at_almost_every_position():
    """The one-hot rows of the adaptive planes rest on 'one symbol has (nearly) every row of a position'.  On the real
    alignment: 91 % of the positions show a single valid symbol, 96.7 % pass the cost model's test for ONE row (other valid
    symbols <= N / 320), all but 0.04 % that for two — 1.03 plane rows per position, like the synthetic model's 1.035."""
    _, _, sequences = load()
    n, positions, beside_first, beside_second = one_hot_statistics(sequences)
    one_row = beside_first <= n / 320
    two_rows = ~one_row & (beside_second <= n / 320)
    print(f"single-symbol positions {(beside_first == 0).mean():.4f}; one row {one_row.mean():.4f}, two rows {two_rows.mean():.4f}, "
          f"more {1 - one_row.mean() - two_rows.mean():.5f}; keys if every position took one row: {beside_first.sum() / (n * positions):.2e} of the cells")
    assert (beside_first == 0).mean() > 0.9 and one_row.mean() > 0.96 and (one_row | two_rows).mean() > 0.999


@pytest.mark.gpu
def test_real_alignment_is_reencoded_into_one_hot_rows(built):
    """The same on the device: the 1 000 real sequences 70 times over (70 000 rows: long enough rows for finalize to
    re-encode) end up with NO plane row at almost every position (the most numerous symbol is derived) — in ONE with
    SILO_GPU_TUNE_COMPACT_INDEX 3 (a row for that symbol too), in 2 code planes with the one-hot rows switched off (2) — and
    answer with 70 times the known counts."""
    from silo_amd import binding
    from silo_amd.engine import Engine

    genomes, lineages, sequences = load()
    copies = 70
    doc = {"nucleotideSequences": [g for g in genomes["nucleotideSequences"] if g["name"] == "main"], "genes": []}
    lib = binding.load_library()
    n, positions, escapes = layout_statistics(sequences)
    _, _, beside_first, _ = one_hot_statistics(sequences)
    for knob in (0, 3, 2):  # the most numerous symbol derived / a one-hot row for it too / code planes only
        with Engine(doc) as engine:
            part = engine.add_partition(copies * len(sequences))
            for k in range(copies):
                engine.append_sequences(part, "main", False, k * len(sequences), sequences)
            previous = lib.silo_gpu_tune(4, knob)
            try:
                engine.finalize()
            finally:
                lib.silo_gpu_tune(4, previous)
            store = engine.partition_store(0)
            rows = int(lib.silo_gpu_store_scan_rows(store.handle, 0, 0, positions))
            keys = int(lib.silo_gpu_store_scan_escapes(store.handle, 0))
            print(f"knob {knob}: plane rows per position {rows / positions:.4f}, escape keys {keys} = {keys / (copies * n * positions):.2e} of the cells")
            if knob == 0:
                assert lib.silo_gpu_store_scan_planes(store.handle, 0) == 0 and rows < 0.08 * positions
                assert keys <= copies * int(beside_first.sum())
            elif knob == 3:
                assert lib.silo_gpu_store_scan_planes(store.handle, 0) == 1 and rows < 1.08 * positions
                assert keys <= copies * int(beside_first.sum())
            else:
                assert lib.silo_gpu_store_scan_planes(store.handle, 0) == 2 and rows < 2.01 * positions
                assert keys <= copies * int(escapes.sum())  # positions that keep their identity planes list nothing
            for (position, symbol), want in KNOWN.items():
                assert engine.execute_query(query(position, symbol)) == [{"count": copies * want}]
            rows = engine.execute_query({"action": {"type": "Mutations", "minProportion": 0.5}, "filterExpression": {"type": "True"}})
            by_name = {row["mutation"]: row["count"] for row in rows}
            assert by_name["C241T"] == copies * 740 and by_name["A23403G"] == copies * 748
