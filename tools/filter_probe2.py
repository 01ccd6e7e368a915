"""Config-2 filter query under rocprofv3: kernel duration of k_filter_eval (run with --kernel-trace --stats)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "lapis-silo_amd")]
import bench  # noqa: E402

engine, model, tree, lineage, window = bench.build_engine(10_000_000, 0, 1, None, 0)
query = bench.filter_query(model, tree).encode()
for _ in range(300):
    engine.execute_text(query)
t0 = time.perf_counter()
for _ in range(300):
    engine.execute_text(query)
print("us per query", (time.perf_counter() - t0) / 300 * 1e6, engine.last_trace())
