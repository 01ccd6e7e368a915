"""SURVEY.md §8(f) row 4, the payload half: sequence stores imported from the reference's own storage form — per position a
portable-format roaring bitmap per symbol with one symbol flipped or deleted (position.h:27-37, position.cpp:42-127) and the
missing symbol row-wise (sequence_store.cpp:153-190) — answer like stores built from the characters.  The payloads are made by
oracle/roaring_format.py from the oracle's restatement of SequenceStorePartition::fill; CRoaring is not in the image and the
reference ships no serialized state, so the FORMAT is pinned only by its published specification: parity unpinned."""
import hashlib
import json
import os
import struct

import numpy as np
import pytest

from oracle import roaring_format as rf
from oracle import silo_oracle as so

HERE = os.path.dirname(os.path.abspath(__file__))
VECTORS = json.load(open(os.path.join(HERE, "golden", "roaring", "vectors.json")))["vectors"]


@pytest.mark.parametrize("vector", VECTORS, ids=lambda v: f"{v['name']}-{'runs' if v['use_runs'] else 'plain'}")
def test_oracle_reader_on_the_committed_vectors(vector):
    ids = rf.deserialize(bytes.fromhex(vector["payload_hex"]))
    assert len(ids) == vector["n"] and ids == sorted(ids)
    assert hashlib.sha256(struct.pack(f"<{len(ids)}I", *ids)).hexdigest() == vector["ids_sha256"]
    assert rf.serialize(ids, use_runs=vector["use_runs"]).hex() == vector["payload_hex"]


def test_malformed_payloads_are_refused_by_the_oracle_reader():
    good = rf.serialize(list(range(10, 5000)) + [70000])
    for bad in (good[:3], b"\x00" * 8, good[:-1]):
        with pytest.raises((ValueError, struct.error)):
            rf.deserialize(bad)


@pytest.mark.gpu
def test_device_expansion_of_the_committed_vectors(built):
    """Every vector as the bitmap of symbol A at position 0 of a 320 000-row store: the expanded plane holds exactly its ids."""
    from silo_amd import binding

    n = 320_000
    reference = np.array([1, 2], dtype=np.uint8)
    for vector in VECTORS:
        payload = bytes.fromhex(vector["payload_hex"])
        ids = np.array(rf.deserialize(payload), dtype=np.int64)
        with binding.GpuStore(n, [dict(name="s", alphabet="nuc", reference=reference)]) as store:
            binding.import_position(store.handle, 0, 0, {1: payload})
            store.finalize()
            want = np.zeros(n, dtype=bool)
            want[ids[ids < n]] = True
            from oracle import dense

            got = dense.unpack_bits(store.plane_download(0, 0, 1), n)
            assert np.array_equal(got, want), vector["name"]
    with binding.GpuStore(1000, [dict(name="s", alphabet="nuc", reference=reference)]) as store:
        with pytest.raises(binding.SiloGpuError):
            binding.import_position(store.handle, 0, 0, {1: b"\x01\x02\x03\x04\x05\x06\x07\x08\x09"})  # unknown cookie
        with pytest.raises(binding.SiloGpuError):
            binding.import_position(store.handle, 0, 0, {1: rf.serialize(range(5000))[:-1]})  # container past the end
        with pytest.raises(binding.SiloGpuError):
            binding.import_position(store.handle, 0, 0, {15: rf.serialize([1])})  # the missing symbol is imported row-wise
        # a row can have one symbol at a position: overlapping bitmaps of one call, and a second import of the same position, are refused
        with pytest.raises(binding.SiloGpuError, match="overlap"):
            binding.import_position(store.handle, 0, 1, {1: rf.serialize([3, 4, 5]), 2: rf.serialize([5, 6])})
        binding.import_position(store.handle, 0, 0, {1: rf.serialize([3, 4, 5]), 2: rf.serialize([6, 7])})
        with pytest.raises(binding.SiloGpuError, match="overlap"):
            binding.import_position(store.handle, 0, 0, {3: rf.serialize([4])})


def position_payloads(position):
    """{symbol: payload} of an oracle Position: every stored (non-empty) bitmap, plus the flipped symbol's even when empty."""
    payloads = {}
    for symbol, bits in position.bitmaps.items():
        if bits or symbol == position.flipped:
            payloads[symbol] = rf.serialize(so.ids_from_bits(bits))
    return payloads


@pytest.mark.gpu
@pytest.mark.parametrize("state", ["deleted", "flipped"])
def test_imported_example_dataset_answers_like_the_appended_one(built, state):
    """testBaseData/exampleDataset through the reference's storage form: the oracle's SequenceStorePartition::fill gives every
    Position as the reference holds it after optimizeBitmaps ('deleted': the most numerous symbol stored empty) or before it
    ('flipped': the reference symbol stored as its complement); serialized, imported on the device, and the reference's own
    e2e goldens (counts and Mutations rows) come out."""
    from silo_amd import binding
    from silo_amd.engine import Engine
    from tests import dataset
    from tests.test_engine_gpu import build_engine

    data = dataset.load_example_dataset()
    genomes = json.load(open(dataset.GOLDEN + "/exampleDataset/reference_genomes.json"))
    engine = Engine(genomes, data["alias"])
    config = dataset.load_database_config()
    engine.set_schema(config["primary_key"], config["date_to_sort_by"])
    n = len(data["keys"])
    part = engine.add_partition(n)
    store = engine.partition_store(part)
    for is_aa, alphabet, sequences, references in (
        (False, so.Nucleotide, data["nuc"], {g["name"]: g["sequence"] for g in genomes["nucleotideSequences"]}),
        (True, so.AminoAcid, data["aa"], {g["name"]: g["sequence"] for g in genomes["genes"]}),
    ):
        for name, rows in sequences.items():
            oracle_store = so.SequenceStorePartition(alphabet, [alphabet.char_to_symbol(c) for c in references[name]])
            if state == "deleted":
                oracle_store.fill(rows)
            else:  # the state between interpret() and optimizeBitmaps: every position still holds its reference symbol flipped
                for begin in range(0, len(rows), oracle_store.BUFFER_SIZE):
                    oracle_store.interpret(rows[begin:begin + oracle_store.BUFFER_SIZE])
            sid = engine.seqstore_id(part, name, is_aa)
            binding.import_missing_rows(store.handle, sid, 0, [rf.serialize(sorted(positions)) for positions in oracle_store.missing_symbol_bitmaps])
            for p, position in enumerate(oracle_store.positions):
                assert (position.deleted is not None) == (state == "deleted") or position.deleted is None
                binding.import_position(store.handle, sid, p, position_payloads(position), position.flipped, position.deleted)
    column_types = {"aa_insertion": "aaInsertion"}
    for column, kind in config["metadata"]:
        engine.append_metadata(part, column, column_types.get(kind, kind), [row.get(column) or "" for row in data["rows"]])
    engine.finalize()
    appended = build_engine(data)
    try:
        for case in dataset.load_query_fixtures("queries"):
            status, document = engine.execute_raw(case["query"])
            assert (status, document) == (200, {"queryResult": case["expectedQueryResult"]}), case["file"]
        for query in (
            {"action": {"type": "Mutations", "minProportion": 0.0}, "filterExpression": {"type": "True"}},
            {"action": {"type": "AminoAcidMutations", "minProportion": 0.0}, "filterExpression": {"type": "True"}},
            {"action": {"type": "FastaAligned", "sequenceName": ["main", "S"], "orderByFields": ["gisaid_epi_isl"]}, "filterExpression": {"type": "True"}},
            {"action": {"type": "Aggregated"}, "filterExpression": {"type": "Maybe", "child": {"type": "NucleotideEquals", "position": 122, "symbol": "A"}}},
        ):
            assert engine.execute_raw(query) == appended.execute_raw(query), query
    finally:
        appended.close()
        engine.close()
