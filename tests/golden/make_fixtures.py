"""Copies the DATA the reference's own tests hold for this path into tests/golden/ (run in the build
container, where /root/reference is mounted; the GPU box has no reference tree).

Nothing here executes or copies reference *code*: inputs are the e2e test data set
(testBaseData/exampleDataset: aligned FASTA files, metadata TSV, reference genomes, lineage aliases),
the 1000-sequence ndjson sample, and the query/expected-result JSON fixtures of
endToEndTests/test/{queries,invalidQueries}.  Sequence files are re-compressed with lzma (stdlib) so
the tests need neither zstd nor the reference tree.
"""
import ctypes
import json
import lzma
import os
import shutil

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))

# filter / action types on the hot path
HOT_FILTERS = {
    "True", "False", "And", "Or", "Not", "N-Of", "Maybe", "Exact", "NucleotideEquals", "AminoAcidEquals",
    "HasNucleotideMutation", "HasAminoAcidMutation", "PangoLineage",
}
HOT_ACTIONS = {"Aggregated", "Mutations", "AminoAcidMutations"}
# SURVEY.md §8(f) "next" rows: metadata predicates feeding the filter tree, Aggregated with groupByFields, Details,
# FastaAligned.  Their fixtures go to queries_next/ and invalidQueries_next/.
NEXT_FILTERS = HOT_FILTERS | {"StringEquals", "IntEquals", "IntBetween", "FloatEquals", "FloatBetween", "DateBetween",
                              "InsertionContains", "AminoAcidInsertionContains"}
NEXT_ACTIONS = HOT_ACTIONS | {"Details", "FastaAligned", "Fasta", "Insertions", "AminoAcidInsertions"}
NEXT_INVALID = {"GroupByLineageInvalidOrderBy.json", "OffsetNegative.json", "insertionContains_empty.json",
                "insertionContains_invalidPattern.json", "insertionContains_invalidPattern2.json", "insertionsAAseparation.json",
                "insertionsInvalidColumn.json", "insertionsInvalidSequence.json"}
HOT_INVALID = {
    "sequencePos0Filter.json", "invalidMutationsMinProportion.json", "nuc_mutations_no_proportion.json",
    "aa_mutations_no_proportion.json", "invalidAction.json",
}


def zstd_decompress(data: bytes) -> bytes:
    lib = ctypes.CDLL("/opt/conda/lib/libzstd.so")
    lib.ZSTD_getFrameContentSize.restype = ctypes.c_ulonglong
    lib.ZSTD_getFrameContentSize.argtypes = [ctypes.c_char_p, ctypes.c_size_t]
    lib.ZSTD_decompress.restype = ctypes.c_size_t
    lib.ZSTD_decompress.argtypes = [ctypes.c_char_p, ctypes.c_size_t, ctypes.c_char_p, ctypes.c_size_t]
    lib.ZSTD_isError.argtypes = [ctypes.c_size_t]
    size = lib.ZSTD_getFrameContentSize(data, len(data))
    if size >= 0xFFFFFFFFFFFFFFFE:  # unknown: grow a buffer
        size = 256 * len(data)
    out = ctypes.create_string_buffer(size)
    n = lib.ZSTD_decompress(out, size, data, len(data))
    if lib.ZSTD_isError(n):
        raise RuntimeError("zstd decompression failed")
    return out.raw[:n]


def filter_types(node, acc):
    if isinstance(node, dict):
        if "type" in node:
            acc.add(node["type"])
        for value in node.values():
            filter_types(value, acc)
    elif isinstance(node, list):
        for value in node:
            filter_types(value, acc)


def main():
    src = os.path.join(REF, "testBaseData", "exampleDataset")
    dst = os.path.join(HERE, "exampleDataset")
    os.makedirs(dst, exist_ok=True)
    for name in ["reference_genomes.json", "pangolineage_alias.json", "small_metadata_set.tsv", "database_config.yaml",
                 "preprocessing_config.yaml"]:
        shutil.copyfile(os.path.join(src, name), os.path.join(dst, name))
    for name in sorted(os.listdir(src)):
        path = os.path.join(src, name)
        if name.endswith(".fasta"):
            raw = open(path, "rb").read()
            base = name
        elif name.endswith(".fasta.zst"):
            raw = zstd_decompress(open(path, "rb").read())
            base = name[: -len(".zst")]
        elif name.endswith(".fasta.xz"):
            raw = lzma.decompress(open(path, "rb").read())
            base = name[: -len(".xz")]
        else:
            continue
        with lzma.open(os.path.join(dst, base + ".xz"), "wb", preset=9) as out:
            out.write(raw)

    # 1000-sequence sample (BASELINE config 1): keep only what the path reads
    sample = zstd_decompress(open(os.path.join(REF, "testBaseData", "exampleDataset1000Sequences", "sample.ndjson.zst"), "rb").read())
    dst1000 = os.path.join(HERE, "exampleDataset1000Sequences")
    os.makedirs(dst1000, exist_ok=True)
    with lzma.open(os.path.join(dst1000, "main_aligned.txt.xz"), "wt", preset=9) as out:
        for line in sample.decode().splitlines():
            if not line.strip():
                continue
            record = json.loads(line)
            seq = record["alignedNucleotideSequences"]["main"]
            lineage = record["metadata"].get("pango_lineage") if "metadata" in record else None
            out.write(f"{lineage or ''}\t{seq if seq is not None else ''}\n")
    shutil.copyfile(
        os.path.join(REF, "testBaseData", "exampleDataset1000Sequences", "reference_genomes.json"),
        os.path.join(dst1000, "reference_genomes.json"),
    )

    qsrc = os.path.join(REF, "endToEndTests", "test", "queries")
    qdst = os.path.join(HERE, "queries")
    os.makedirs(qdst, exist_ok=True)
    ndst = os.path.join(HERE, "queries_next")
    os.makedirs(ndst, exist_ok=True)
    kept, kept_next = [], []
    for name in sorted(os.listdir(qsrc)):
        case = json.load(open(os.path.join(qsrc, name)))
        types = set()
        filter_types(case["query"].get("filterExpression"), types)
        action = case["query"]["action"]
        if types <= HOT_FILTERS and action.get("type") in HOT_ACTIONS and not action.get("groupByFields"):
            shutil.copyfile(os.path.join(qsrc, name), os.path.join(qdst, name))
            kept.append(name)
        elif types <= NEXT_FILTERS and action.get("type") in NEXT_ACTIONS:
            shutil.copyfile(os.path.join(qsrc, name), os.path.join(ndst, name))
            kept_next.append(name)
    isrc = os.path.join(REF, "endToEndTests", "test", "invalidQueries")
    idst = os.path.join(HERE, "invalidQueries")
    os.makedirs(idst, exist_ok=True)
    for name in sorted(HOT_INVALID):
        shutil.copyfile(os.path.join(isrc, name), os.path.join(idst, name))
    os.makedirs(os.path.join(HERE, "invalidQueries_next"), exist_ok=True)
    for name in sorted(NEXT_INVALID):
        shutil.copyfile(os.path.join(isrc, name), os.path.join(HERE, "invalidQueries_next", name))
    print(f"kept {len(kept)} query fixtures:", " ".join(kept))
    print(f"kept {len(kept_next)} next-row query fixtures:", " ".join(kept_next))


# The four scenarios of src/silo/preprocessing/preprocessor.test.cpp:31-88 (input directories of testBaseData plus the
# expected sequence count, query and result, transcribed from the test as data).
FASTA_ALIGNED_QUERY = {"action": {"type": "FastaAligned", "sequenceName": ["someShortGene", "secondSegment"], "orderByFields": ["accessionVersion"]},
                       "filterExpression": {"type": "True"}}
FASTA_ALIGNED_EXPECTED = [
    {"accessionVersion": "1.1", "someShortGene": "MADS", "secondSegment": "NNNNNNNNNNNNNNNN"},
    {"accessionVersion": "1.3", "someShortGene": "XXXX", "secondSegment": "NNNNNNNNNNNNNNNN"},
]
GROUP_QUERY = {"action": {"type": "Aggregated", "groupByFields": ["group"], "orderByFields": ["group"]}, "filterExpression": {"type": "True"}}
GROUP_EXPECTED = [{"count": 1, "group": None}, {"count": 1, "group": "dummyValue"}]
PREPROCESSING_SCENARIOS = {
    "fastaFilesWithMissingSequences": (2, FASTA_ALIGNED_QUERY, FASTA_ALIGNED_EXPECTED),
    "ndjsonWithNullSequences": (2, FASTA_ALIGNED_QUERY, FASTA_ALIGNED_EXPECTED),
    "ndjsonWithSqlKeywordField": (2, GROUP_QUERY, GROUP_EXPECTED),
    "tsvWithSqlKeywordField": (2, GROUP_QUERY, GROUP_EXPECTED),
}


def copy_preprocessing_scenarios():
    for name, (count, query, expected) in PREPROCESSING_SCENARIOS.items():
        src = os.path.join(REF, "testBaseData", name)
        dst = os.path.join(HERE, "preprocessing", name)
        os.makedirs(dst, exist_ok=True)
        for file_name in sorted(os.listdir(src)):
            shutil.copyfile(os.path.join(src, file_name), os.path.join(dst, file_name))
        json.dump({"cite": "src/silo/preprocessing/preprocessor.test.cpp:31-88", "expectedSequenceCount": count, "query": query,
                   "expectedQueryResult": expected}, open(os.path.join(dst, "expected.json"), "w"), indent=1)


if __name__ == "__main__":
    copy_preprocessing_scenarios()
    main()
