"""K5 (column predicate) and K6 (group-by histogram) at full size: time per launch and effective bandwidth."""
import argparse
import ctypes
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "lapis-silo_amd")]
from silo_amd import binding, synth  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--sequences", type=int, default=10_000_000)
ap.add_argument("--reps", type=int, default=20)
args = ap.parse_args()
n = args.sequences
rng = np.random.default_rng(1)
lib = binding.load_library()
store = binding.GpuStore(n, [dict(name="s", alphabet="nuc", reference=np.ones(4, dtype=np.uint8))])
tree = synth.make_lineage_tree(2000)
lineage = synth.assign_lineages(n, tree, 7).astype(np.uint32)        # Zipf-skewed dictionary ids
country = rng.integers(0, 50, size=n).astype(np.uint32)
age = rng.integers(0, 100, size=n).astype(np.int32)
qc = rng.random(n)
filt = store.bitset_alloc()
store.bitset_upload(filt, np.packbits(rng.random(store.row_words * 64) < 0.4, bitorder="little").view(np.uint64))
start, stop = binding.GpuEvent(), binding.GpuEvent()


def timed(label, nbytes, launch):
    for _ in range(2):
        launch()
    start.record()
    for _ in range(args.reps):
        launch()
    stop.record()
    ms = start.elapsed_ms(stop) / args.reps
    print(f"{label:45s} {ms * 1e3:8.1f} us  {nbytes / ms / 1e6:7.0f} GB/s")


out = store.bitset_alloc()
for label, column, dtype in (("compare int32 >= 30", age, np.int32), ("compare uint32 == 7", lineage, np.uint32), ("compare float64 < 0.5", qc, np.float64)):
    pointer = store.upload_column(column)
    scalar = np.array([30 if dtype == np.int32 else (7 if dtype == np.uint32 else 0.5)], dtype=dtype)
    comparator = {np.int32: 3, np.uint32: 0, np.float64: 2}[dtype]
    value_type = binding.GpuStore.VALUE_TYPES[np.dtype(dtype)]
    timed(label, column.nbytes + n // 8,
          lambda: binding._check(lib.silo_gpu_bitset_from_compare(store.handle, out, pointer, value_type, comparator, scalar.ctypes.data_as(ctypes.c_void_p), None)))
    store.free(pointer)

lineage_dev, country_dev = store.upload_column(lineage), store.upload_column(country)
for label, pointers, cards in (("group by lineage (2000 groups, skewed, LDS)", [lineage_dev], [2000]),
                               ("group by country (50 groups, LDS)", [country_dev], [50]),
                               ("group by lineage x country (100000, global)", [lineage_dev, country_dev], [2000, 50])):
    n_bins = int(np.prod(cards))
    counts = store.malloc(4 * n_bins)
    ids = (ctypes.c_void_p * len(pointers))(*[p.value for p in pointers])
    cardinalities = (ctypes.c_uint32 * len(cards))(*cards)
    for f, suffix in ((filt, ", 40 % filter"), (None, ", all rows")):
        timed(label + suffix, 4 * n * len(pointers) + n // 8,
              lambda: binding._check(lib.silo_gpu_group_count(store.handle, f, ids, cardinalities, len(pointers), counts, None)))
    store.free(counts)
