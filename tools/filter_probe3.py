import os, sys, time
ROOT = "/root/repo"
sys.path[:0] = [ROOT, os.path.join(ROOT, "lapis-silo_amd")]
import bench
from silo_amd import binding
lib = binding.load_library()
for knob in (-1, 0, -1, 0):
    lib.silo_gpu_tune(4, knob)
    engine, model, tree, lineage, window = bench.build_engine(10_000_000, 0, 1, None, 0)
    query = bench.filter_query(model, tree).encode()
    for _ in range(500):
        engine.execute_text(query)
    best = 1e9
    for rep in range(5):
        t0 = time.perf_counter()
        for _ in range(500):
            engine.execute_text(query)
        best = min(best, (time.perf_counter() - t0) / 500 * 1e6)
    print("compact index", "on" if knob == 0 else "off", "HBM GB", engine.partition_store(0).device_bytes / 1e9, "us per query (best of 5)", best, engine.last_trace(), flush=True)
    engine.close()
