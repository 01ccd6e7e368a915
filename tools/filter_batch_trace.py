#!/usr/bin/env python3
"""Where the host time of a batch of filter -> Aggregated queries goes: phase marks of silo_engine_execute_batch.
usage: filter_batch_trace.py [sequences] [queries]"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "lapis-silo_amd")]
import bench  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
q = int(sys.argv[2]) if len(sys.argv) > 2 else 64
engine, model, tree, lineage, window = bench.build_engine(n, 0, 1, None, 0)
batch = [bench.filter_query(model, tree, k).encode() for k in range(q)]
engine.execute_batch_text(batch)
for _ in range(3):
    t0 = time.perf_counter()
    engine.execute_batch_text(batch)
    wall = (time.perf_counter() - t0) * 1e6
    print(f"wall {wall:.0f} us for {q} queries;", json.dumps(engine.last_trace()), flush=True)
t0 = time.perf_counter()
for wire in batch:
    engine.execute_text(wire)
print(f"one by one: {(time.perf_counter() - t0) * 1e6 / q:.1f} us per query;", json.dumps(engine.last_trace()))

import threading

for clients in (1, 2, 4, 8, 16):
    done = []

    def client():
        k = 0
        end = time.perf_counter() + 1.0
        while time.perf_counter() < end:
            engine.execute_batch_text(batch)
            k += 1
        done.append(k)

    threads = [threading.Thread(target=client) for _ in range(clients)]
    t0 = time.perf_counter()
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    print(f"{clients:2d} client threads x batches of {q}: {sum(done) * q / (time.perf_counter() - t0):9.0f} queries/s", flush=True)
