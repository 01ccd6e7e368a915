#!/usr/bin/env python3
"""How many sequences one GPU holds: a nucleotide genome of N sequences built in two passes (counted, then written straight
into the adaptive planes: no build-time planes), the bench query on it.  usage: capacity_probe.py [sequences] [reps] [auto]"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "lapis-silo_amd")]
import bench  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 25_000_000
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
forced = not (len(sys.argv) > 3 and sys.argv[3] == "auto")  # "auto": leave the choice of one or two passes to the engine (free memory)
t0 = time.time()
engine, model, tree, lineage, window = bench.build_engine(n, 0, 1, None, 0, two_pass=forced)
store = engine.partition_store(0)
print(f"{n} sequences x {model.positions} positions built in {time.time() - t0:.1f} s (two passes of the generator), {store.device_bytes / 1e9:.1f} GB on the device", flush=True)
member = tree.subtree(tree.names.index(bench.QUERY_LINEAGE))
want = int(member[lineage].sum())
count = engine.execute_query({"action": {"type": "Aggregated"}, "filterExpression": json.loads(bench.make_query())["filterExpression"]})
assert count == [{"count": want}], (count, want)
query = bench.make_query().encode()
for _ in range(2):
    status, body = engine.execute_text(query)
assert status == 200
t0 = time.perf_counter()
for _ in range(reps):
    engine.execute_text(query)
seconds = (time.perf_counter() - t0) / reps
rows = json.loads(body)["queryResult"]
print(f"Mutations query: {seconds * 1e3:.2f} ms = {n * model.positions / seconds:.3e} positions*sequences/s, {len(rows)} rows; filter count {want}", flush=True)
engine.close()
