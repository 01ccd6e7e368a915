"""Prints the per-phase trace of a Mutations query and of the config-2 filter query (host overhead analysis)."""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "lapis-silo_amd")]
import bench  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--sequences", type=int, default=1_000_000)
args = ap.parse_args()
engine, model, tree, lineage, window = bench.build_engine(args.sequences, 0, 1, None, 0)
for name, query in (("mutations", bench.make_query()), ("filter", bench.filter_query(model, tree))):
    for _ in range(5):
        engine.execute_query(query)
    traces = []
    walls = []
    for _ in range(20):
        t0 = time.perf_counter()
        engine.execute_raw(query)
        walls.append((time.perf_counter() - t0) * 1e6)
        traces.append(engine.last_trace())
    keys = list(traces[0].keys())
    median = {k: sorted(t[k] for t in traces)[len(traces) // 2] for k in keys}
    print(name, "python wall µs (median):", sorted(walls)[len(walls) // 2], "trace µs:", json.dumps(median))
