"""K7 (k_reconstruct_sequences) at scale: time to rebuild the aligned sequences of R random rows of a big store."""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "lapis-silo_amd")]
import bench  # noqa: E402
from oracle import synth as oracle_synth  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--sequences", type=int, default=10_000_000)
args = ap.parse_args()
engine, model, tree, lineage, window = bench.build_engine(args.sequences, 0, 1, None, 0)
view = engine.partition_store(0)
lib = bench.__dict__.get("binding") or __import__("silo_amd.binding", fromlist=["x"])
binding = lib
glib = binding.load_library()


class Store:
    """The slice of GpuStore the probe needs, over the engine's store handle."""

    def reconstruct_sequences(self, seqstore_id, rows):
        import ctypes

        rows = np.ascontiguousarray(rows, dtype=np.uint32)
        positions = model.positions
        rows_dev, out = ctypes.c_void_p(), ctypes.c_void_p()
        binding._check(glib.silo_gpu_upload_column(rows.ctypes.data_as(ctypes.c_void_p), len(rows), 1, ctypes.byref(rows_dev)))
        binding._check(glib.silo_gpu_malloc(len(rows) * positions, ctypes.byref(out)))
        start, stop = binding.GpuEvent(), binding.GpuEvent()
        start.record()
        binding._check(glib.silo_gpu_reconstruct_sequences(view.handle, seqstore_id, rows_dev, len(rows), out, None))
        stop.record()
        self.kernel_ms = start.elapsed_ms(stop)
        host = np.empty(len(rows) * positions, dtype=np.uint8)
        binding._check(glib.silo_gpu_memcpy_d2h(host.ctypes.data_as(ctypes.c_void_p), out, host.nbytes, None))
        glib.silo_gpu_free(rows_dev)
        glib.silo_gpu_free(out)
        return host.reshape(len(rows), positions)


store = Store()
rng = np.random.default_rng(3)
chars = np.frombuffer(b"-ACGTRYSWKMBDHVN", dtype=np.uint8)
for n_rows in (10, 100, 1000, 10000):
    rows = np.sort(rng.choice(args.sequences, size=n_rows, replace=False)).astype(np.uint32)
    store.reconstruct_sequences(0, rows[:2])
    t0 = time.perf_counter()
    got = store.reconstruct_sequences(0, rows)
    elapsed = time.perf_counter() - t0
    check = rows[:: max(1, n_rows // 5)][:5]
    want = chars[oracle_synth.symbol_matrix(model, check.astype(np.int64), np.arange(model.positions))]
    ok = np.array_equal(got[:: max(1, n_rows // 5)][:5], want)
    print(f"{n_rows:6d} rows x {model.positions} positions: {elapsed * 1e3:9.2f} ms incl. upload / download, kernel {store.kernel_ms:8.2f} ms ({n_rows * model.positions / elapsed / 1e9:.2f} G cells/s), matches the generator: {ok}")
