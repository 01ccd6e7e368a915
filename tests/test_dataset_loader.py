"""The C++ reader of the reference's input formats (lapis-silo_amd/host/dataset_loader.cpp): the e2e goldens
again, this time with the engine built by silo_engine_create_from_directory from the files the reference's own
preprocessing reads (metadata TSV + FASTA.xz, and an ndjson.zst export of the same rows)."""
import ctypes
import json
import lzma
import os
import shutil
import subprocess

import pytest

from tests import dataset

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXAMPLE = os.path.join(dataset.GOLDEN, "exampleDataset")


def zstd_compress(data: bytes) -> bytes:
    lib = ctypes.CDLL("libzstd.so.1")
    lib.ZSTD_compressBound.restype = ctypes.c_size_t
    lib.ZSTD_compressBound.argtypes = [ctypes.c_size_t]
    lib.ZSTD_compress.restype = ctypes.c_size_t
    lib.ZSTD_compress.argtypes = [ctypes.c_char_p, ctypes.c_size_t, ctypes.c_char_p, ctypes.c_size_t, ctypes.c_int]
    bound = lib.ZSTD_compressBound(len(data))
    out = ctypes.create_string_buffer(bound)
    n = lib.ZSTD_compress(out, bound, data, len(data), 3)
    return out.raw[:n]


def write_ndjson_dataset(directory, compression):
    """The example data set as the reference's ndjson input (preprocessor.cpp:87-131), two rows with null genomes."""
    data = dataset.load_example_dataset()
    os.makedirs(directory, exist_ok=True)
    for name in ("reference_genomes.json", "pangolineage_alias.json", "database_config.yaml"):
        shutil.copyfile(os.path.join(EXAMPLE, name), os.path.join(directory, name))
    def insertion_map(text, names, default):
        """'25701:CCC,S:214:EPE' -> {sequence: ['position:insertion', ...]}: insertions travel as top-level maps in the
        ndjson format (metadata_info.cpp:61-94), not as metadata fields."""
        out = {name: [] for name in names}
        for entry in (text.split(",") if text else []):
            parts = entry.split(":")
            if len(parts) == 2:
                out[default].append(entry)
            else:
                out[parts[0]].append(":".join(parts[1:]))
        return out

    lines = []
    for i, row in enumerate(data["rows"]):
        record = {
            "metadata": {k: (v if v != "" else None) for k, v in row.items() if k not in ("nucleotideInsertions", "aminoAcidInsertions")},
            "alignedNucleotideSequences": {name: seqs[i] for name, seqs in data["nuc"].items()},
            "alignedAminoAcidSequences": {name: seqs[i] for name, seqs in data["aa"].items()},
            "unalignedNucleotideSequences": {name: data["unaligned"].get(name, [None] * len(data["rows"]))[i] for name in data["nuc"]},
            "nucleotideInsertions": insertion_map(row["nucleotideInsertions"], data["nuc"], "main"),
            "aminoAcidInsertions": insertion_map(row["aminoAcidInsertions"], data["aa"], None),
        }
        lines.append(json.dumps(record))
    text = ("\n".join(lines) + "\n").encode()
    if compression == "zst":
        name = "input.ndjson.zst"
        payload = zstd_compress(text)
    elif compression == "xz":
        name = "input.ndjson.xz"
        payload = lzma.compress(text)
    else:
        name = "input.ndjson"
        payload = text
    open(os.path.join(directory, name), "wb").write(payload)
    open(os.path.join(directory, "preprocessing_config.yaml"), "w").write(
        f'ndjsonInputFilename: "{name}"\npangoLineageDefinitionFilename: "pangolineage_alias.json"\nreferenceGenomeFilename: "reference_genomes.json"\n')


def run_goldens(engine):
    for case in dataset.load_query_fixtures("queries"):
        status, document = engine.execute_raw(case["query"])
        assert status == 200, (case["file"], document)
        assert document == {"queryResult": case["expectedQueryResult"]}, case["file"]
    for case in dataset.load_query_fixtures("invalidQueries") + dataset.load_query_fixtures("invalidQueries_next"):
        status, document = engine.execute_raw(case["query"])
        assert (status, document) == (400, case["expectedError"]), case["file"]

    def execute(query):
        status, document = engine.execute_raw(query)
        assert status == 200, document
        return document["queryResult"]

    for case in dataset.load_query_fixtures("queries_next"):  # metadata columns read from the TSV / ndjson
        dataset.check_next_row_case(case, execute)


@pytest.mark.gpu
def test_tsv_and_fasta_directory(built):
    from silo_amd.engine import Engine

    with Engine.from_directory(EXAMPLE) as engine:
        assert engine.summary == {"sequenceCount": 100, "nucleotideStores": 2, "aminoAcidStores": 12, "lineageColumns": 1, "nullSequences": 0}
        run_goldens(engine)


@pytest.mark.gpu
@pytest.mark.parametrize("source", ["tsv", "ndjson"])
def test_rows_lie_in_the_reference_order(built, tmp_path, source):
    """The loader lays the rows out as the reference's preprocessing does (preprocessor.cpp:159-227, database_config.cpp:190-198):
    by partitionBy key (pango_lineage), then dateToSortBy (date), then primaryKey — so a lineage is a row range and the dates
    ascend inside it.  Seen through the engine: the bitset of a lineage filter is ONE run of rows, Details (which walks the
    rows in storage order) returns them in that order; `sortRows: false` keeps the file's order (and the same answers)."""
    import numpy as np

    from silo_amd.engine import Engine

    directory = EXAMPLE
    if source == "ndjson":
        write_ndjson_dataset(str(tmp_path), "none")
        directory = str(tmp_path)
    details = {"action": {"type": "Details", "fields": ["pango_lineage", "date", "gisaid_epi_isl"]}, "filterExpression": {"type": "True"}}
    with Engine.from_directory(directory) as engine:
        rows = engine.execute_query(details)
        assert len(rows) == 100
        key = lambda row: (row["pango_lineage"] or "", row["date"] is None, row["date"] or "", row["gisaid_epi_isl"])
        assert [key(row) for row in rows] == sorted(key(row) for row in rows)
        assert len({row["pango_lineage"] for row in rows}) > 5
        words, count = engine.evaluate_filter({"type": "PangoLineage", "column": "pango_lineage", "value": "B.1.1.7", "includeSublineages": False})
        selected = np.nonzero(np.unpackbits(words.view(np.uint8), bitorder="little"))[0]
        assert count == len(selected) == 48 and selected[-1] - selected[0] == 47  # one run of rows
        sorted_counts = engine.execute_query({"action": {"type": "Mutations", "minProportion": 0.3}, "filterExpression": {"type": "True"}})
    if source == "tsv":
        for name in os.listdir(EXAMPLE):
            origin = os.path.join(EXAMPLE, name)
            if name == "preprocessing_config.yaml":
                with open(origin) as handle, open(os.path.join(str(tmp_path), name), "w") as out:
                    out.write(handle.read().rstrip("\n") + "\nsortRows: false\n")
            else:
                os.symlink(origin, os.path.join(str(tmp_path), name))
        with Engine.from_directory(str(tmp_path)) as engine:
            unsorted = engine.execute_query(details)
            assert [key(row) for row in unsorted] != sorted(key(row) for row in unsorted) and sorted(map(key, unsorted)) == [key(row) for row in rows]
            assert engine.execute_query({"action": {"type": "Mutations", "minProportion": 0.3}, "filterExpression": {"type": "True"}}) == sorted_counts


@pytest.mark.gpu
def test_tsv_and_fasta_directory_in_two_passes(built, tmp_path):
    """twoPassBuild: true in preprocessing_config.yaml: every sequence store of the directory is fed twice (counted, then
    written by silo_gpu_store_build_pass's second pass) — the same summary, the same goldens."""
    from silo_amd.engine import Engine

    for name in os.listdir(EXAMPLE):
        source = os.path.join(EXAMPLE, name)
        if name == "preprocessing_config.yaml":
            with open(source) as handle, open(os.path.join(str(tmp_path), name), "w") as out:
                out.write(handle.read().rstrip("\n") + "\ntwoPassBuild: true\n")
        else:
            os.symlink(source, os.path.join(str(tmp_path), name))
    with Engine.from_directory(str(tmp_path)) as engine:
        assert engine.summary == {"sequenceCount": 100, "nucleotideStores": 2, "aminoAcidStores": 12, "lineageColumns": 1, "nullSequences": 0}
        run_goldens(engine)


@pytest.mark.gpu
@pytest.mark.parametrize("compression", ["zst", "xz", "none"])
def test_ndjson_directory(built, tmp_path, compression):
    from silo_amd.engine import Engine

    write_ndjson_dataset(str(tmp_path), compression)
    with Engine.from_directory(str(tmp_path)) as engine:
        assert engine.summary["sequenceCount"] == 100
        run_goldens(engine)


@pytest.mark.gpu
def test_ndjson_directory_in_two_passes(built, tmp_path):
    """twoPassBuild: true with ndjson input: the file is read twice — the aligned sequences counted, then every record handled
    as usual with the sequences written by the second pass (null genomes included) — the same summary, the same goldens."""
    from silo_amd.engine import Engine

    write_ndjson_dataset(str(tmp_path), "zst")
    with open(os.path.join(str(tmp_path), "preprocessing_config.yaml"), "a") as config:
        config.write("twoPassBuild: true\n")
    with Engine.from_directory(str(tmp_path)) as engine:
        assert engine.summary["sequenceCount"] == 100
        two_pass_summary = dict(engine.summary)
        run_goldens(engine)
    write_ndjson_dataset(str(tmp_path), "zst")
    with Engine.from_directory(str(tmp_path)) as engine:
        assert engine.summary == two_pass_summary  # (null sequences are not counted twice)


@pytest.mark.gpu
@pytest.mark.parametrize("scenario", ["fastaFilesWithMissingSequences", "ndjsonWithNullSequences", "ndjsonWithSqlKeywordField", "tsvWithSqlKeywordField"])
def test_reference_preprocessing_scenarios(built, scenario):
    """src/silo/preprocessing/preprocessor.test.cpp:31-130: the reference's own input directories (missing segments and
    genes, null sequences, a column named like an SQL keyword) with the sequence count and query result it expects."""
    from silo_amd.engine import Engine

    directory = os.path.join(dataset.GOLDEN, "preprocessing", scenario)
    expected = json.load(open(os.path.join(directory, "expected.json")))
    with Engine.from_directory(directory) as engine:
        assert engine.summary["sequenceCount"] == expected["expectedSequenceCount"]
        status, document = engine.execute_raw(expected["query"])
        assert status == 200, document
        assert document["queryResult"] == expected["expectedQueryResult"]


@pytest.mark.gpu
def test_cli_answers_queries(built):
    next_rows = [case for case in dataset.load_query_fixtures("queries_next")
                 if case["file"] in ("GroupByDivision.json", "dateBetween.json", "insertionsAction.json", "fastaAligned_multiple.json", "DetailsOrderBy.json")]
    assert len(next_rows) == 5
    cases = dataset.load_query_fixtures("queries")[:6] + next_rows + dataset.load_query_fixtures("invalidQueries")[:2]
    stdin = "".join(json.dumps(case["query"]) + "\n" for case in cases)
    proc = subprocess.run([os.path.join(ROOT, "lapis-silo_amd", "lib", "silo_query"), EXAMPLE], input=stdin, capture_output=True, text=True, timeout=600)
    assert proc.returncode == 0, proc.stderr[-2000:]
    lines = proc.stdout.strip().splitlines()
    assert len(lines) == len(cases)
    for case, line in zip(cases, lines):
        status, _, body = line.partition("\t")
        if "expectedQueryResult" in case:
            assert status == "200" and json.loads(body) == {"queryResult": case["expectedQueryResult"]}
        else:
            assert status == "400" and json.loads(body) == case["expectedError"]
    assert "Execution (action)" in proc.stderr


def test_loader_reports_problems_not_crashes(built, tmp_path):
    """CPU: the loader's host-side parsing and error reporting (no device is reached before the store is created)."""
    from silo_amd import engine

    lib = engine.load_library()

    def load(directory):
        handle, summary = ctypes.c_void_p(), ctypes.c_void_p()
        rc = lib.silo_engine_create_from_directory(str(directory).encode(), 0, ctypes.byref(handle), ctypes.byref(summary))
        return rc, lib.silo_engine_last_error().decode()

    rc, message = load(tmp_path / "does-not-exist")
    assert rc != 0 and "cannot open" in message
    both = tmp_path / "both"
    both.mkdir()
    shutil.copyfile(os.path.join(EXAMPLE, "database_config.yaml"), both / "database_config.yaml")
    (both / "preprocessing_config.yaml").write_text('ndjsonInputFilename: "a.ndjson"\nmetadataFilename: "b.tsv"\n')
    rc, message = load(both)
    assert rc != 0 and message == "Cannot specify both a ndjsonInputFilename ('a.ndjson') and metadataFilename('b.tsv')."


@pytest.mark.gpu
def test_overridden_file_prefixes(built, tmp_path):
    """preprocessing_config_reader.test.cpp:35-50 (test_preprocessing_config_with_overridden_defaults.yaml): a
    genePrefix of its own and an EMPTY nucleotideSequencePrefix — the sequence files are then `<name>.fasta` — give the
    same database as the default names."""
    import shutil

    from silo_amd.engine import Engine

    source = os.path.join(dataset.GOLDEN, "preprocessing", "tsvWithSqlKeywordField")
    expected = json.load(open(os.path.join(source, "expected.json")))
    for name in os.listdir(source):
        target = name
        if name.startswith("gene_"):
            target = "aaSeq_" + name[len("gene_"):]
        elif name.startswith("nuc_"):
            target = name[len("nuc_"):]
        shutil.copy(os.path.join(source, name), os.path.join(str(tmp_path), target))
    with open(os.path.join(str(tmp_path), "preprocessing_config.yaml"), "a") as out:
        out.write('\ngenePrefix: "aaSeq_"\nnucleotideSequencePrefix: ""\n')
    with Engine.from_directory(str(tmp_path)) as engine:
        assert engine.summary["sequenceCount"] == expected["expectedSequenceCount"]
        status, document = engine.execute_raw(expected["query"])
        assert status == 200 and document["queryResult"] == expected["expectedQueryResult"]


@pytest.mark.gpu
@pytest.mark.parametrize("input_file,references,message", [
    # sequence_info.test.cpp:40-52 — the fixture is not valid JSON (a comma is missing before its extra "testSecondSequence"), so
    # what the reference's test sees is DuckDB refusing the file; the loader refuses it with its parse error
    ("oneline_second_nuc.json.zst", "exampleDataset1000Sequences", "parse error"),
    ("oneline_without_ORF.json.zst", "exampleDataset1000Sequences", "which is contained in the reference sequences is not contained in the input file"),  # :54-66
])
def test_sequence_names_of_the_first_record_are_validated(built, tmp_path, input_file, references, message):
    """SequenceInfo::validate (sequence_info.cpp:91-157) on the reference's own one-line ndjson fixtures: a record with
    a sequence the reference genomes do not know, and one without a gene they do know, are refused with its messages."""
    import shutil

    from silo_amd.engine import Engine, SiloEngineError

    directory = str(tmp_path)
    shutil.copy(os.path.join(dataset.GOLDEN, "ndjsonFiles", input_file), os.path.join(directory, input_file))
    shutil.copy(os.path.join(dataset.GOLDEN, references, "reference_genomes.json"), os.path.join(directory, "reference_genomes.json"))
    with open(os.path.join(directory, "preprocessing_config.yaml"), "w") as out:
        out.write(f'ndjsonInputFilename: "{input_file}"\nreferenceGenomeFilename: "reference_genomes.json"\n')
    with open(os.path.join(directory, "database_config.yaml"), "w") as out:
        out.write("schema:\n  instanceName: test\n  metadata:\n    - name: strain\n      type: string\n  primaryKey: strain\n")
    with pytest.raises(SiloEngineError, match=message):
        Engine.from_directory(directory)


@pytest.mark.gpu
def test_a_sequence_unknown_to_the_reference_genomes_is_refused(built, tmp_path):
    """The other direction of SequenceInfo::validate (sequence_info.cpp:106-116), with a well-formed record: the
    exampleDataset as ndjson holds "testSecondSequence", which the reference genomes of the 1000-sequence data set lack."""
    import shutil

    from silo_amd.engine import Engine, SiloEngineError

    write_ndjson_dataset(str(tmp_path), "none")
    shutil.copy(os.path.join(dataset.GOLDEN, "exampleDataset1000Sequences", "reference_genomes.json"), os.path.join(str(tmp_path), "reference_genomes.json"))
    with pytest.raises(SiloEngineError, match="The aligned nucleotide sequence testSecondSequence which is contained in the input file .* is not contained in the reference sequences"):
        Engine.from_directory(str(tmp_path))


@pytest.mark.gpu
@pytest.mark.parametrize("ndjson", [False, True])
def test_a_configured_column_missing_from_the_input_is_refused(built, tmp_path, ndjson):
    """metadata_info.test.cpp:8-32 (and its ndjson twin): a database config that names a column the metadata file /
    the first record does not hold."""
    import shutil

    from silo_amd.engine import Engine, SiloEngineError

    directory = str(tmp_path)
    if ndjson:
        write_ndjson_dataset(directory, "none")
    else:
        for name in os.listdir(EXAMPLE):
            shutil.copy(os.path.join(EXAMPLE, name), os.path.join(directory, name))
    config = open(os.path.join(directory, "database_config.yaml")).read()
    config = config.replace("  metadata:\n", "  metadata:\n    - name: notInMetadata\n      type: pango_lineage\n", 1)
    open(os.path.join(directory, "database_config.yaml"), "w").write(config)
    with pytest.raises(SiloEngineError, match="The metadata field 'notInMetadata' which is contained in the database config is not contained in the input"):
        Engine.from_directory(directory)
