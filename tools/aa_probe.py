"""Amino-acid leg of BASELINE.json configs[3]: per-gene scan time (HIP events), all genes back to back, and the
phase trace of the whole AminoAcidMutations query — where the time of a 12-gene query goes."""
import argparse
import ctypes
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "lapis-silo_amd")]
import bench  # noqa: E402
from silo_amd import binding  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--sequences", type=int, default=1_000_000)
ap.add_argument("--reps", type=int, default=20)
ap.add_argument("--nuc-positions", type=int, default=None, help="cut the nucleotide store to a stub (10 M sequences: the genes alone are 73 GB)")
ap.add_argument("--variants", type=str, default="0", help="SILO_GPU_TUNE_SCAN_VARIANT values to try (16 = 2 words per thread)")
ap.add_argument("--rows", type=str, default="", help="positions per block to try for the one-call scan of all genes, e.g. 12,24,32,64")
args = ap.parse_args()

engine, model, tree, lineage, window = bench.build_engine(args.sequences, 0, 1, None, 0, with_genes=True, nuc_positions=args.nuc_positions)
lib = binding.load_library()
store = engine.partition_store(0)
genes = bench.load_reference_genomes(True)["genes"]
member = tree.subtree(tree.names.index(bench.QUERY_LINEAGE))
filt = ctypes.c_void_p()
binding._check(lib.silo_gpu_bitset_alloc(store.handle, ctypes.byref(filt)))
binding._check(lib.silo_gpu_bitset_from_lineages(store.handle, filt, member.ctypes.data_as(ctypes.c_void_p), len(member), None))
w8 = 8 * ((args.sequences + 63) // 64)

ids, lengths, tables = [], [], []
for gene in genes:
    ids.append(engine.seqstore_id(0, gene["name"], True))
    lengths.append(len(gene["sequence"]))
    table = ctypes.c_void_p()
    binding._check(lib.silo_gpu_malloc(4 * lengths[-1] * 22, ctypes.byref(table)))
    binding._check(lib.silo_gpu_memset_async(table, 0, 4 * lengths[-1] * 22, None))
    tables.append(table)

print("scan planes per position:", {gene["name"]: (int(lib.silo_gpu_store_scan_planes(store.handle, sid)), int(lib.silo_gpu_store_scan_escapes(store.handle, sid)))
                                    for gene, sid in zip(genes, ids)}, flush=True)
start, stop = binding.GpuEvent(), binding.GpuEvent()
total_single = 0.0
for gene, sid, length, table in zip(genes, ids, lengths, tables):
    for _ in range(2):
        binding._check(lib.silo_gpu_mutations_scan(store.handle, sid, filt, 0, length, table, None))
    start.record()
    for _ in range(args.reps):
        binding._check(lib.silo_gpu_mutations_scan(store.handle, sid, filt, 0, length, table, None))
    stop.record()
    ms = start.elapsed_ms(stop) / args.reps
    total_single += ms
    nbytes = length * 22 * w8
    print(f"{gene['name']:6s} P={length:5d} rows={length * 22:6d} {ms * 1e3:8.1f} us  {nbytes / ms / 1e6:7.0f} GB/s  ({lib.silo_gpu_last_scan_kernel().decode()})")
all_bytes = sum(lengths) * 22 * w8
print(f"sum of per-gene times {total_single:.3f} ms -> {all_bytes / total_single / 1e6:.0f} GB/s")

start.record()
for _ in range(args.reps):
    for sid, length, table in zip(ids, lengths, tables):
        binding._check(lib.silo_gpu_mutations_scan(store.handle, sid, filt, 0, length, table, None))
stop.record()
ms = start.elapsed_ms(stop) / args.reps
print(f"12 launches back to back {ms:.3f} ms -> {all_bytes / ms / 1e6:.0f} GB/s")

ranges = (ctypes.c_uint32 * (3 * len(ids)))(*[v for sid, length in zip(ids, lengths) for v in (sid, 0, length)])
table_array = (ctypes.c_void_p * len(ids))(*[t.value for t in tables])
filters = (ctypes.c_void_p * 1)(filt.value)
for variant, rows in [(int(v), r) for v in args.variants.split(",") for r in [0] + [int(r) for r in args.rows.split(",") if r]]:
    lib.silo_gpu_tune(0, rows)
    lib.silo_gpu_tune(1, variant)
    for _ in range(2):
        binding._check(lib.silo_gpu_mutations_scan_ranges(store.handle, ranges, len(ids), filters, 1, table_array, None))
    start.record()
    for _ in range(args.reps):
        binding._check(lib.silo_gpu_mutations_scan_ranges(store.handle, ranges, len(ids), filters, 1, table_array, None))
    stop.record()
    ms = start.elapsed_ms(stop) / args.reps
    phys = sum(lengths) * 5 * w8
    print(f"one call over 12 genes, variant {variant}, positions per block {rows or 'default'}: {ms:.3f} ms -> {all_bytes / ms / 1e6:.0f} GB/s algorithmic, "
          f"{phys / ms / 1e6:.0f} GB/s physical", flush=True)
lib.silo_gpu_tune(0, 0)
lib.silo_gpu_tune(1, 0)

query = json.dumps({"action": {"type": "AminoAcidMutations", "minProportion": 0.05},
                    "filterExpression": json.loads(bench.make_query())["filterExpression"]}).encode()
for _ in range(5):
    engine.execute_text(query)
walls, traces = [], []
for _ in range(args.reps):
    t0 = time.perf_counter()
    engine.execute_text(query)
    walls.append((time.perf_counter() - t0) * 1e6)
    traces.append(engine.last_trace())
keys = list(traces[0].keys())
median = {k: sorted(t[k] for t in traces)[len(traces) // 2] for k in keys}
print("whole query wall us (median):", sorted(walls)[len(walls) // 2], "trace us:", json.dumps(median))
