"""Batches of 2, 3, 4, 8 Mutations queries at 10 M rows: the escape pass with 2 / 4 / 8 filters per block as the batch size
says (variant 0); variant 45 — always the eight-filter kernel with its packed sums — was a temporary knob, see profiles/r03_notes.md."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "lapis-silo_amd")]
import bench  # noqa: E402
from silo_amd import binding  # noqa: E402

lib = binding.load_library()
engine, model, tree, lineage, window = bench.build_engine(10_000_000, 0, 1, None, 0, with_genes=True)
names = ("B.1", "B.2", "B.3", "B.1.1", "B.1.2", "B.1.3", "B.2.1", "B.2.2")
for action in ("Mutations", "AminoAcidMutations"):
    for size in (2, 3, 4, 8):
        batch = [json.dumps({"action": {"type": action, "minProportion": 0.05},
                             "filterExpression": {"type": "PangoLineage", "column": "pango_lineage", "value": name, "includeSublineages": True}}).encode()
                 for name in names[:size]]
        line = f"{action:20s} batch of {size}:"
        for variant in (0, 45, 0, 45):
            lib.silo_gpu_tune(1, variant)
            for _ in range(3):
                engine.execute_batch_text(batch)
            t0 = time.perf_counter()
            for _ in range(30):
                engine.execute_batch_text(batch)
            line += f"  v{variant} {(time.perf_counter() - t0) / 30 * 1e3:.3f} ms"
        print(line, flush=True)
lib.silo_gpu_tune(1, 0)
