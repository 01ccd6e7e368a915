"""Config-2 filter query (32 leaves -> Aggregated): latency per query for the K3 leaf-batch variants."""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "lapis-silo_amd")]
import bench  # noqa: E402
from silo_amd import binding  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--sequences", type=int, default=10_000_000)
args = ap.parse_args()
engine, model, tree, lineage, window = bench.build_engine(args.sequences, 0, 1, None, 0)
lib = binding.load_library()
query = bench.filter_query(model, tree).encode()
for batch in (8, 16, 8, 16):
    lib.silo_gpu_tune(2, batch)
    for _ in range(50):
        engine.execute_text(query)
    t0 = time.perf_counter()
    n = 2000
    for _ in range(n):
        engine.execute_text(query)
    wall = (time.perf_counter() - t0) / n * 1e6
    print(f"leaf batch {batch:2d}: {wall:7.1f} us per query  trace {engine.last_trace()}")
