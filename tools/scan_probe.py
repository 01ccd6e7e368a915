"""Quick K1 probe: synthetic nucleotide store, time the Mutations scan with HIP events."""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "lapis-silo_amd")]
from silo_amd import binding, synth  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--sequences", type=int, default=1_000_000)
ap.add_argument("--positions", type=int, default=29903)
ap.add_argument("--lineages", type=int, default=2000)
ap.add_argument("--rows", type=str, default="32,64,128")
ap.add_argument("--reps", type=int, default=5)
ap.add_argument("--variants", type=str, default="0,10,12")
ap.add_argument("--alphabet", type=str, default="nuc")
ap.add_argument("--lineage-order", action="store_true", help="rows laid out lineage by lineage, sublineages behind their parent")
ap.add_argument("--keycost", type=str, default="0", help="SILO_GPU_TUNE_KEY_COST values to try (the store is rebuilt for each)")
ap.add_argument("--side", type=str, default="0", help="SILO_GPU_TUNE_SIDE_STREAM values to try (escape pass: 0 low-priority side stream, 1 default priority, 2 caller's stream)")
args = ap.parse_args()

n, positions = args.sequences, args.positions
t0 = time.time()
tree = synth.make_lineage_tree(args.lineages)
lineage = synth.assign_lineages(n, tree, synth.DEFAULT_SEED)
if args.lineage_order:
    rank_of = np.empty(len(tree.names), dtype=np.int64)
    rank_of[sorted(range(len(tree.names)), key=lambda k: [int(part) for part in tree.names[k].split(".")[1:]])] = np.arange(len(tree.names))
    lineage = lineage[np.argsort(rank_of[lineage], kind="stable")]
ref = synth.random_reference(positions, "nuc", 1)
model = synth.make_model(n, ref, "nuc", tree, lineage)
print(f"model built in {time.time() - t0:.1f}s", flush=True)
ap_alphabet = "aa" if args.alphabet == "aa" else "nuc"
if ap_alphabet == "aa":
    ref = synth.random_reference(positions, "aa", 1)
    model = synth.make_model(n, ref, "aa", tree, lineage)
member = tree.subtree(1)
w8 = 8 * ((n + 63) // 64)
n_sym = 5 if ap_alphabet == "nuc" else 22
alg_bytes = positions * n_sym * w8 + w8
reference_counts = None
for keycost in [int(k) for k in args.keycost.split(",")]:
    t0 = time.time()
    store = binding.GpuStore(n, [dict(name="main", alphabet=ap_alphabet, reference=ref)])
    store.tune(6, keycost)
    store.generate_synthetic(0, model)
    store.finalize()
    store.tune(6, 0)
    fptr = store.bitset_alloc()
    store.bitset_from_lineages(fptr, member)
    counts = store.malloc(4 * positions * n_sym)
    start, stop = binding.GpuEvent(), binding.GpuEvent()
    print(f"keycost {keycost}: built in {time.time() - t0:.1f}s, {store.device_bytes / 1e9:.1f} GB, filter cardinality {store.popcount(fptr)} of {n}, "
          f"plane rows {store.scan_rows(0, 0, positions)}, escape keys {store.scan_escapes(0)}", flush=True)
    for variant, rows, side in [(int(v), int(r), int(s)) for v in args.variants.split(",") for r in args.rows.split(",") for s in args.side.split(",")]:
        store.tune(1, variant)
        store.tune(0, rows)
        store.tune(5, side)
        best = 1e9
        for rep in range(args.reps + 1):
            store.memset(counts, 0, 4 * positions * n_sym)
            start.record()
            store.mutations_scan_async(0, fptr, 0, positions, counts)
            stop.record()
            ms = start.elapsed_ms(stop)
            if rep > 0:
                best = min(best, ms)
        c = store.read(counts, np.uint32, positions * n_sym)
        if reference_counts is None:
            reference_counts = c.copy()
        assert np.array_equal(c, reference_counts), "variant changed the counts"
        physical = store.scan_rows(0, 0, positions) * w8 + w8 + 8 * store.scan_escapes(0)
        print(f"  variant={variant:2d} rows_per_block={rows:5d} side={side}  {best:8.3f} ms  physical {physical / best / 1e6:7.1f} GB/s  "
              f"{n * positions / best * 1e3:.3e} pos*seq/s  kernel={store.last_scan_kernel()}", flush=True)
    store.tune(0, 0); store.tune(1, 0); store.tune(5, 0)
    store.close()
print("checksum", int(reference_counts.astype(np.uint64).sum()))
