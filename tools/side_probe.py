"""The escape-key pass and the passes for derived symbols on side streams (SILO_GPU_TUNE_SIDE_STREAM 0, 1) or all on the
caller's stream (2): the headline query, the 12-gene amino-acid query, 8 batched Mutations queries, at 10 M rows."""
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
nuc = bench.make_query().encode()
aa = json.dumps({"action": {"type": "AminoAcidMutations", "minProportion": 0.05},
                 "filterExpression": {"type": "PangoLineage", "column": "pango_lineage", "value": bench.QUERY_LINEAGE, "includeSublineages": True}}).encode()
batch = [json.dumps({"action": {"type": "Mutations", "minProportion": 0.05},
                     "filterExpression": {"type": "PangoLineage", "column": "pango_lineage", "value": name, "includeSublineages": True}}).encode()
         for name in ("B.1", "B.2", "B.3", "B.1.1", "B.1.2", "B.1.3", "B.2.1", "B.2.2")]


def timed(call, reps):
    for _ in range(5):
        call()
    t0 = time.perf_counter()
    for _ in range(reps):
        call()
    return (time.perf_counter() - t0) / reps * 1e3


for mode in (0, 1, 2, 0, 1, 2):
    lib.silo_gpu_tune(5, mode)
    print(f"side mode {mode}: nucleotide {timed(lambda: engine.execute_text(nuc), 200):.3f} ms   amino acid {timed(lambda: engine.execute_text(aa), 100):.3f} ms   "
          f"8 batched {timed(lambda: engine.execute_batch_text(batch), 50):.3f} ms", flush=True)
lib.silo_gpu_tune(5, 0)
