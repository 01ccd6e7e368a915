"""configs[2] filter -> Aggregated at 10 M rows from native request threads (silo_engine_run_clients): 1, 2, 4, 8, 16 clients."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "lapis-silo_amd")]
import bench  # noqa: E402

engine, model, tree, lineage, window = bench.build_engine(10_000_000, 0, 1, None, 0)
query = bench.filter_query(model, tree).encode()
for _ in range(100):
    engine.execute_text(query)
for clients in (1, 2, 4, 8, 16, 1, 8):
    per_second, body = engine.run_clients(query, clients, 2.0)
    print(f"{clients:2d} clients: {per_second:9.0f} queries/s  ({clients / per_second * 1e6:6.1f} us per query and client)  {body[:60]}", flush=True)
