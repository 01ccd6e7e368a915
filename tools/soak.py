"""Concurrency soak: 8 client threads hammer one engine with a fixed mix of queries (filters, Mutations, metadata
predicates, group-by, batches); every response must equal the one computed sequentially up front, and device memory must
not creep."""
import argparse
import ctypes
import json
import os
import random
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "lapis-silo_amd")]
import bench  # noqa: E402
from silo_amd import binding  # noqa: E402
bench.binding = binding

ap = argparse.ArgumentParser()
ap.add_argument("--sequences", type=int, default=1_000_000)
ap.add_argument("--seconds", type=float, default=60.0)
ap.add_argument("--threads", type=int, default=8)
ap.add_argument("--side", type=int, default=0, help="SILO_GPU_TUNE_SIDE_STREAM (escape pass: 0 side stream, 2 caller's stream, 3 position-major keys)")
ap.add_argument("--batch-share", type=float, default=0.05)
args = ap.parse_args()
bench.binding.load_library().silo_gpu_tune(5, args.side)

engine, model, tree, lineage, window = bench.build_engine(args.sequences, 0, 1, None, 0, with_genes=True, with_metadata=True)
rng = random.Random(1)
lineages = ["B.1", "B.2", "B.3", "B.1.1", "B.1.2", "B.2.1", "B.3.3", "B.1.1.1"]


def lineage_filter(name):
    return {"type": "PangoLineage", "column": "pango_lineage", "value": name, "includeSublineages": True}


queries = [bench.filter_query(model, tree)]
for name in lineages:
    queries.append(json.dumps({"action": {"type": "Mutations", "minProportion": 0.05}, "filterExpression": lineage_filter(name)}))
    queries.append(json.dumps({"action": {"type": "Aggregated"}, "filterExpression": {"type": "And", "children": [
        lineage_filter(name), {"type": "StringEquals", "column": "country", "value": f"C{rng.randint(0, 49)}"},
        {"type": "IntBetween", "column": "age", "from": rng.randint(0, 40), "to": rng.randint(41, 99)}]}}))
    queries.append(json.dumps({"action": {"type": "Aggregated", "groupByFields": ["country"], "orderByFields": ["country"]},
                               "filterExpression": lineage_filter(name)}))
queries.append(json.dumps({"action": {"type": "AminoAcidMutations", "minProportion": 0.1, "sequenceName": ["S", "N"]},
                           "filterExpression": {"type": "Not", "child": lineage_filter("B.1")}}))
queries.append(json.dumps({"action": {"type": "Aggregated", "groupByFields": ["country", "age"], "orderByFields": ["count", "country", "age"], "limit": 20},
                           "filterExpression": {"type": "Maybe", "child": {"type": "NucleotideEquals", "position": 241, "symbol": "T"}}}))
# selective filters: single deep lineages (a few hundred rows) go through the sparse-filter gather (K1s), whose scratch
# blocks are pooled across the client threads
deep = sorted((name for name in tree.names if name.count(".") >= 5), key=len)[-6:]
for name in deep:
    exact = {"type": "PangoLineage", "column": "pango_lineage", "value": name, "includeSublineages": False}
    queries.append(json.dumps({"action": {"type": "Mutations", "minProportion": 0.05}, "filterExpression": exact}))
    queries.append(json.dumps({"action": {"type": "AminoAcidMutations", "minProportion": 0.05}, "filterExpression": exact}))
queries = [q.encode() for q in queries]
expected = [engine.execute_text(q) for q in queries]
assert all(status == 200 for status, _ in expected), [body for status, body in expected if status != 200][:1]
batch = queries[1:13:3][:4] + [queries[0]]
expected_batch = engine.execute_batch_text(batch)
assert expected_batch == [expected[queries.index(q)] for q in batch]
store = engine.partition_store(0)
lib = bench.binding.load_library() if hasattr(bench, "binding") else None

errors = []
done = [0] * args.threads
stop_at = time.perf_counter() + args.seconds


def client(index):
    local = random.Random(100 + index)
    while time.perf_counter() < stop_at and not errors:
        if local.random() < args.batch_share:
            got = engine.execute_batch_text(batch)
            if got != expected_batch:
                wrong = [k for k, (a, b) in enumerate(zip(got, expected_batch)) if a != b]
                detail = ""
                if wrong:
                    a, b = json.loads(got[wrong[0]][1]), json.loads(expected_batch[wrong[0]][1])
                    rows_a, rows_b = a.get("queryResult", a), b.get("queryResult", b)
                    differing = [(x, y) for x, y in zip(rows_a, rows_b) if x != y][:3] if isinstance(rows_a, list) else (a, b)
                    detail = f"rows {len(rows_a) if isinstance(rows_a, list) else '-'} vs {len(rows_b) if isinstance(rows_b, list) else '-'}; first differences {differing}"
                errors.append(("batch", index, wrong, detail))
        else:
            k = local.randrange(len(queries))
            if engine.execute_text(queries[k]) != expected[k]:
                errors.append((k, index))
        done[index] += 1


threads = [threading.Thread(target=client, args=(i,)) for i in range(args.threads)]
t0 = time.perf_counter()
for t in threads:
    t.start()
for t in threads:
    t.join()
elapsed = time.perf_counter() - t0
print(f"{sum(done)} queries from {args.threads} threads in {elapsed:.1f} s ({sum(done) / elapsed:.0f}/s), errors: {errors[:5]}")
sys.exit(1 if errors else 0)
