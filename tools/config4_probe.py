#!/usr/bin/env python3
"""Phase marks of the configs[4] shard batch (bench.config4_workload): where the time of the 200-query batch goes."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "lapis-silo_amd")]
import bench  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 6_250_000
engine, _, tree, _, _ = bench.build_engine(n, 0, 1, None, 0, with_genes=True)
reference_text = bench.load_reference_genomes(False)["nucleotideSequences"][0]["sequence"]
genes = bench.load_reference_genomes(True)["genes"]
out = bench.config4_workload(engine, tree, reference_text, 29903, sum(len(g["sequence"]) for g in genes), n, 1, lambda: None)
print(json.dumps({k: v for k, v in out.items() if k != "workload"}))
print(json.dumps(engine.last_trace()))
engine.close()
