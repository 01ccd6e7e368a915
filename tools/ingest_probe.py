"""Times B1 (k_transpose_sequences via silo_gpu_store_append_sequences): aligned characters -> bit planes."""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "lapis-silo_amd")]
from silo_amd import binding  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--sequences", type=int, default=65536)
ap.add_argument("--positions", type=int, default=29903)
ap.add_argument("--batch", type=int, default=16384)
args = ap.parse_args()
rng = np.random.default_rng(1)
chars = np.frombuffer(b"ACGT-N", dtype=np.uint8)
ref = np.ones(args.positions, dtype=np.uint8)
batch = chars[rng.choice(6, size=(args.batch, args.positions), p=[0.3, 0.2, 0.2, 0.27, 0.02, 0.01])]
with binding.GpuStore(args.sequences, [dict(name="main", alphabet="nuc", reference=ref)]) as store:
    t0 = time.perf_counter()
    done = 0
    while done < args.sequences:
        n = min(args.batch, args.sequences - done)
        store.append_sequences(0, done, batch[:n])
        done += n
    store.synchronize()
    dt = time.perf_counter() - t0
    store.finalize()
    cells = args.sequences * args.positions
    print(f"{args.sequences} x {args.positions}: {dt:.3f} s, {cells / dt / 1e9:.2f} G cells/s, {cells / dt / 1e9:.2f} GB/s of characters")
    counts = store.mutations_scan(0)
    print("check", counts[:2].tolist(), int(counts.sum()), "expected non-N cells ~", int(cells * 0.99))
