#!/usr/bin/env python3
"""What one rank of a position-sharded Mutations query costs without the collective: the bench query on a database that
holds only 1/ranks of the genome's positions.  usage: shard_probe.py [sequences] [ranks] [reps] [shard]"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "lapis-silo_amd")]
import bench  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
ranks = int(sys.argv[2]) if len(sys.argv) > 2 else 8
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 20
only_shard = len(sys.argv) > 4 and sys.argv[4] == "shard"
for share in ((ranks,) if only_shard else (ranks, 1)):
    positions = (29903 + share - 1) // share
    engine, model, tree, lineage, window = bench.build_engine(n, 0, 1, None, 0, nuc_positions=positions)
    query = bench.make_query().encode()
    for _ in range(3):
        engine.execute_text(query)
    walls, traces = [], []
    for _ in range(reps):
        t0 = time.perf_counter()
        engine.execute_text(query)
        walls.append((time.perf_counter() - t0) * 1e3)
        traces.append(engine.last_trace())
    median = {k: sorted(t[k] for t in traces)[len(traces) // 2] for k in traces[0]}
    print(f"{positions} positions (1/{share} of the genome): {sorted(walls)[len(walls) // 2]:.3f} ms per query (median of {reps}); phase marks µs: {json.dumps(median)}", flush=True)
    engine.close()
