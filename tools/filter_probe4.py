"""Filter -> Aggregated one by one at 10 M rows: end-to-end latency by the shape of the tree (1, 8, 32 leaves; the configs[2] tree)."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "lapis-silo_amd")]
import bench  # noqa: E402

engine, model, tree, lineage, window = bench.build_engine(10_000_000, 0, 1, None, 0)
full = json.loads(bench.filter_query(model, tree))
leaves = []
for child in full["filterExpression"]["children"]:
    node = child.get("child", child)
    leaves += node["children"]
shapes = {
    "1 leaf": leaves[0],
    "Or of 8": {"type": "Or", "children": leaves[:8]},
    "Or of 32": {"type": "Or", "children": leaves},
    "And of 4 Or of 8": {"type": "And", "children": [{"type": "Or", "children": leaves[8 * k:8 * k + 8]} for k in range(4)]},
    "configs[2]": full["filterExpression"],
}
from silo_amd import binding  # noqa: E402
lib = binding.load_library()
for variant, name, expression in [(v, n, e) for v in (0, 41) for n, e in shapes.items()]:
    lib.silo_gpu_tune(1, variant)
    query = json.dumps({"action": {"type": "Aggregated"}, "filterExpression": expression}).encode()
    for _ in range(300):
        engine.execute_text(query)
    expected = engine.execute_text(query)
    times = []
    for _ in range(2000):
        t0 = time.perf_counter()
        got = engine.execute_text(query)
        times.append(time.perf_counter() - t0)
        assert got == expected
    times.sort()
    print(f"variant {variant:2d} {name:18s} median {times[1000] * 1e6:6.1f} us  mean {sum(times) / 2000 * 1e6:6.1f}  p99 {times[1980] * 1e6:7.1f}  max {times[-1] * 1e6:9.1f}  "
          f"slower than 1 ms: {sum(t > 1e-3 for t in times)}  {engine.last_trace()}", flush=True)
