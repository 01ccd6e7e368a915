"""Input-format reader at a size where its speed shows: N synthetic sequences written as ndjson (plain and .zst) and as
TSV + FASTA, then loaded through silo_engine_create_from_directory; reports MB/s of input text."""
import argparse
import ctypes
import json
import os
import shutil
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "lapis-silo_amd")]
from oracle import synth as oracle_synth  # noqa: E402
from silo_amd import alphabet, synth  # noqa: E402
from silo_amd.engine import Engine  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--sequences", type=int, default=20000)
args = ap.parse_args()
n = args.sequences
genomes = json.load(open(os.path.join(ROOT, "tests", "golden", "exampleDataset", "reference_genomes.json")))
genomes = {"nucleotideSequences": [g for g in genomes["nucleotideSequences"] if g["name"] == "main"], "genes": []}
reference = np.array([alphabet.NUCLEOTIDE.char_to_symbol[c] for c in genomes["nucleotideSequences"][0]["sequence"]], dtype=np.uint8)
tree = synth.make_lineage_tree(200)
lineage = synth.assign_lineages(n, tree, 5)
model = synth.make_model(n, reference, "nuc", tree, lineage, seed=5)
chars = np.frombuffer(b"-ACGTRYSWKMBDHVN", dtype=np.uint8)
root = tempfile.mkdtemp(prefix="loader_probe_")
try:
    config = "schema:\n  instanceName: probe\n  metadata:\n    - name: key\n      type: string\n    - name: lineage\n      type: pango_lineage\n    - name: age\n      type: int\n  primaryKey: key\n"
    for kind in ("ndjson", "tsv"):
        directory = os.path.join(root, kind)
        os.makedirs(directory)
        open(os.path.join(directory, "database_config.yaml"), "w").write(config)
        json.dump(genomes, open(os.path.join(directory, "reference_genomes.json"), "w"))
    nd = open(os.path.join(root, "ndjson", "input.ndjson"), "w")
    tsv = open(os.path.join(root, "tsv", "metadata.tsv"), "w")
    fasta = open(os.path.join(root, "tsv", "nuc_main.fasta"), "w")
    tsv.write("key\tlineage\tage\n")
    for begin in range(0, n, 1000):
        rows = np.arange(begin, min(n, begin + 1000))
        block = chars[oracle_synth.symbol_matrix(model, rows, np.arange(model.positions))]
        for i, row in zip(rows, block):
            sequence = bytes(row).decode()
            nd.write(json.dumps({"metadata": {"key": f"k{i}", "lineage": tree.names[lineage[i]], "age": int(i % 90)},
                                 "alignedNucleotideSequences": {"main": sequence}, "alignedAminoAcidSequences": {},
                                 "unalignedNucleotideSequences": {"main": None}, "nucleotideInsertions": {"main": []}, "aminoAcidInsertions": {}}) + "\n")
            tsv.write(f"k{i}\t{tree.names[lineage[i]]}\t{i % 90}\n")
            fasta.write(f">k{i}\n{sequence}\n")
    for handle in (nd, tsv, fasta):
        handle.close()
    open(os.path.join(root, "ndjson", "preprocessing_config.yaml"), "w").write('ndjsonInputFilename: "input.ndjson"\n')
    open(os.path.join(root, "tsv", "preprocessing_config.yaml"), "w").write('metadataFilename: "metadata.tsv"\n')
    for kind, files in (("ndjson", ["input.ndjson"]), ("tsv", ["metadata.tsv", "nuc_main.fasta"])):
        nbytes = sum(os.path.getsize(os.path.join(root, kind, name)) for name in files)
        t0 = time.perf_counter()
        with Engine.from_directory(os.path.join(root, kind)) as engine:
            elapsed = time.perf_counter() - t0
            count = engine.execute_query({"action": {"type": "Aggregated"}, "filterExpression": {"type": "True"}})[0]["count"]
        print(f"{kind:6s}: {n} sequences, {nbytes / 1e6:.0f} MB of text loaded in {elapsed:.2f} s = {nbytes / elapsed / 1e6:.0f} MB/s ({n / elapsed:.0f} sequences/s), count {count}")
finally:
    shutil.rmtree(root, ignore_errors=True)
