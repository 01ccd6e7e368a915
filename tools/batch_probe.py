"""K1c probe: time Q concurrent Mutations scans as Q single passes vs one batched pass."""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "lapis-silo_amd")]
from silo_amd import binding, synth  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--sequences", type=int, default=1_000_000)
ap.add_argument("--positions", type=int, default=29903)
args = ap.parse_args()
n, positions = args.sequences, args.positions
tree = synth.make_lineage_tree(2000)
lineage = synth.assign_lineages(n, tree, synth.DEFAULT_SEED)
ref = synth.random_reference(positions, "nuc", 1)
model = synth.make_model(n, ref, "nuc", tree, lineage)
store = binding.GpuStore(n, [dict(name="main", alphabet="nuc", reference=ref)])
store.generate_synthetic(0, model)
store.finalize()
filters = []
for root in (1, 2, 3, 4, 5, 6, 7, 8):
    ptr = store.bitset_alloc()
    store.bitset_from_lineages(ptr, tree.subtree(root))
    filters.append(ptr)
import ctypes
lib = store.lib
counts = [store.malloc(4 * positions * 5) for _ in filters]
w8 = 8 * ((n + 63) // 64)
start, stop = binding.GpuEvent(), binding.GpuEvent()
for q in (1, 2, 4, 5, 6, 8):
    fa = (ctypes.c_void_p * q)(*[f.value for f in filters[:q]])
    ca = (ctypes.c_void_p * q)(*[c.value for c in counts[:q]])
    best = 1e9
    for rep in range(4):
        start.record()
        binding._check(lib.silo_gpu_mutations_scan_batch(store.handle, 0, fa, q, 0, positions, ca, None))
        stop.record()
        ms = start.elapsed_ms(stop)
        if rep:
            best = min(best, ms)
    print(f"Q={q}: {best:8.3f} ms per batch = {best / q:7.3f} ms per query; planes stream at {positions * 5 * w8 * -(-q // 8) / best / 1e6:7.1f} GB/s; "
          f"{q * n * positions / best / 1e9 * 1e3:.3e} pos*seq/s aggregate  [{store.last_scan_kernel()}]", flush=True)
