"""K1s probe: time the Mutations scan under filters of decreasing selectivity, gather routing on vs off."""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "lapis-silo_amd")]
from silo_amd import binding, synth  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--sequences", type=int, default=10_000_000)
ap.add_argument("--positions", type=int, default=29903)
ap.add_argument("--reps", type=int, default=5)
ap.add_argument("--divisors", type=str, default="-1,2,4,8,16,32")
args = ap.parse_args()

n, positions = args.sequences, args.positions
tree = synth.make_lineage_tree(2000)
lineage = synth.assign_lineages(n, tree, synth.DEFAULT_SEED)
ref = synth.random_reference(positions, "nuc", 1)
model = synth.make_model(n, ref, "nuc", tree, lineage)
store = binding.GpuStore(n, [dict(name="main", alphabet="nuc", reference=ref)])
t0 = time.time()
store.generate_synthetic(0, model)
store.finalize()
print(f"store {store.device_bytes / 1e9:.1f} GB generated in {time.time() - t0:.1f}s, row_words {store.row_words}", flush=True)

rng = np.random.default_rng(5)
n_words = (n + 63) // 64
filters = {}
for k in (1, 100, 1000, 5000, 20000, 100000):
    words = np.zeros(store.row_words, dtype=np.uint64)
    rows = rng.choice(n, size=k, replace=False)
    np.bitwise_or.at(words, rows // 64, np.uint64(1) << (rows % 64).astype(np.uint64))
    filters[f"{k} random rows"] = words
for k in (10000, 100000, 1000000, 3000000, 5000000):
    words = np.zeros(store.row_words, dtype=np.uint64)
    first = n // 3 // 64
    words[first:first + k // 64] = np.uint64(0xFFFFFFFFFFFFFFFF)
    filters[f"{k} contiguous rows"] = words

counts = store.malloc(4 * positions * 5)
start, stop = binding.GpuEvent(), binding.GpuEvent()
fptr = store.bitset_alloc()
print(f"{'filter':>24} {'sectors':>14} " + " ".join(f"{'div ' + d:>10}" for d in args.divisors.split(",")), flush=True)
for name, words in filters.items():
    store.bitset_upload(fptr, words)
    line = f"{name:>24} {int(np.count_nonzero(words.reshape(-1, 8).any(axis=1))):>14} "
    expected = None
    for divisor in [int(d) for d in args.divisors.split(",")]:
        store.tune(3, divisor)
        best = 1e9
        for rep in range(args.reps + 1):
            store.memset(counts, 0, 4 * positions * 5)
            start.record()
            store.mutations_scan_async(0, fptr, 0, positions, counts)
            stop.record()
            ms = start.elapsed_ms(stop)
            if rep > 0:
                best = min(best, ms)
        got = store.read(counts, np.uint32, positions * 5)
        if expected is None:
            expected = got.copy()
        assert np.array_equal(got, expected), "routing changed the counts"
        line += f" {best:9.3f}ms"
    print(line, flush=True)
store.tune(3, 0)
