#!/usr/bin/env python3
"""The AminoAcidMutations query over all 12 genes (BASELINE.json configs[3], amino-acid leg) on the one-GPU database, for
`rocprofv3 --kernel-trace`: layout of every gene, then the query `reps` times.  usage: aa_profile.py [sequences] [reps] [nucleotide stub positions] [escape-pass modes]"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "lapis-silo_amd")]
import bench  # noqa: E402
from silo_amd import binding  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
nuc_stub = int(sys.argv[3]) if len(sys.argv) > 3 else None
side_modes = [int(v) for v in sys.argv[4].split(",")] if len(sys.argv) > 4 else [0]  # SILO_GPU_TUNE_SIDE_STREAM values to time
lib = binding.load_library()
t0 = time.time()
engine, model, tree, lineage, window = bench.build_engine(n, 0, 1, None, 0, with_genes=True, nuc_positions=nuc_stub)
store = engine.partition_store(0)
print(f"built in {time.time() - t0:.1f} s, {store.device_bytes / 1e9:.1f} GB", flush=True)
for gene in bench.load_reference_genomes(True)["genes"]:
    sid = engine.seqstore_id(0, gene["name"], True)
    length = len(gene["sequence"])
    rows = int(lib.silo_gpu_store_scan_rows(store.handle, sid, 0, length))
    print(f"{gene['name']:6s} P={length:5d} rows/position {rows / length:.2f} escapes {int(lib.silo_gpu_store_scan_escapes(store.handle, sid))}")
query = json.dumps({"action": {"type": "AminoAcidMutations", "minProportion": 0.05}, "filterExpression": json.loads(bench.make_query())["filterExpression"]}).encode()
for side in side_modes:
    lib.silo_gpu_tune(5, side)
    engine.execute_text(query)
    t0 = time.perf_counter()
    for _ in range(reps):
        engine.execute_text(query)
    print(f"AminoAcidMutations (escape pass mode {side}): {(time.perf_counter() - t0) / reps * 1e3:.3f} ms per query", flush=True)
    print(json.dumps(engine.last_trace()))
lib.silo_gpu_tune(5, 0)
