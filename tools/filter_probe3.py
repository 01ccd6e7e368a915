"""Config-2 filter query one by one: end-to-end latency for the leaf-load batch sizes of K3 (SILO_GPU_TUNE_EVAL_LEAF_BATCH)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "lapis-silo_amd")]
import bench  # noqa: E402
from silo_amd import binding  # noqa: E402

engine, model, tree, lineage, window = bench.build_engine(10_000_000, 0, 1, None, 0)
query = bench.filter_query(model, tree).encode()
lib = binding.load_library()
for knob in (0, 16, 0, 16):
    lib.silo_gpu_tune(2, knob)
    for _ in range(300):
        engine.execute_text(query)
    t0 = time.perf_counter()
    for _ in range(1000):
        engine.execute_text(query)
    print("leaf batch", knob, "us per query", (time.perf_counter() - t0) / 1000 * 1e6, engine.last_trace(), flush=True)
