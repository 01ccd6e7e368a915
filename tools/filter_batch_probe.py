#!/usr/bin/env python3
"""Probe of k_filter_eval_batch: Q programs of the configs[2] shape (32 leaf columns each) over synthetic leaf planes,
with the planes (a) in ONE device allocation and (b) in one allocation per plane, timed with HIP events.
usage: filter_batch_probe.py [sequences] [programs]"""
import ctypes
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "lapis-silo_amd")]
from silo_amd import binding as b  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
q_count = int(sys.argv[2]) if len(sys.argv) > 2 else 64
lib = b.load_library()
store = b.GpuStore(n, [dict(name="s", alphabet="nuc", reference=np.ones(2, dtype=np.uint8))])
row_bytes = store.row_words * 8
code = (b.encode(b.OP_OR_N, 0, imm=0 | (8 << 16))
        + b.encode(b.OP_ZERO, 2) + b.encode(b.OP_ZERO, 3) + b.encode(b.OP_ZERO, 4) + b.encode(b.OP_ZERO, 5)
        + b.encode(b.OP_CNT_ADD_N, 2, 0, 4, imm=8 | (8 << 16)) + b.encode(b.OP_CNT_GE, 1, 2, 4, imm=3) + b.encode(b.OP_AND, 0, 0, 1)
        + b.encode(b.OP_OR_N, 1, imm=16 | (8 << 16)) + b.encode(b.OP_ANDNOT, 0, 0, 1)
        + b.encode(b.OP_AND_N, 1, imm=24 | (8 << 16)) + b.encode(b.OP_AND, 0, 0, 1))
simple = b.encode(b.OP_OR_N, 0, imm=0 | (32 << 16))


def run(label, pointers, program_code, slots):
    programs = b.PreparedPrograms([(program_code, pointers[32 * q:32 * q + 32], slots) for q in range(q_count)])
    programs.launch(store.handle)
    start, stop = b.GpuEvent(), b.GpuEvent()
    reps = 10
    start.record()
    for _ in range(reps):
        programs.launch(store.handle)
    stop.record()
    ms = start.elapsed_ms(stop) / reps
    gb = q_count * 32 * (n + 63) // 64 * 8 / 1e9
    print(f"{label}: {ms:.3f} ms per launch (+ table upload, count copy), {gb / ms * 1e3:.0f} GB/s", flush=True)


slab = ctypes.c_void_p()
b._check(lib.silo_gpu_malloc(q_count * 32 * row_bytes, ctypes.byref(slab)))
b._check(lib.silo_gpu_memset_async(slab, 0x5A, q_count * 32 * row_bytes, None))
slab_pointers = [slab.value + k * row_bytes for k in range(q_count * 32)]
run("one slab, configs[2] program", slab_pointers, code, 6)
run("one slab, OR_N over 32 leaves", slab_pointers, simple, 1)
single = []
for k in range(q_count * 32):
    p = ctypes.c_void_p()
    b._check(lib.silo_gpu_malloc(row_bytes, ctypes.byref(p)))
    b._check(lib.silo_gpu_memset_async(p, 0x5A, row_bytes, None))
    single.append(p.value)
run("one allocation per plane, configs[2] program", single, code, 6)
run("one allocation per plane, OR_N over 32 leaves", single, simple, 1)

# aggregate rate of the bare C call from several host threads (each its own stream and its own disjoint programs)
import threading
import time

streams = []
for _ in range(8):
    stream = ctypes.c_void_p()
    b._check(lib.silo_gpu_stream_create(ctypes.byref(stream)))
    streams.append(stream)
for n_threads in (1, 2, 4, 8):
    prepared = [b.PreparedPrograms([(code, slab_pointers[32 * q:32 * q + 32], 6) for q in range(q_count)]) for _ in range(n_threads)]
    done = []

    def worker(k):
        count = 0
        end = time.perf_counter() + 1.0
        while time.perf_counter() < end:
            prepared[k].launch(store.handle, streams[k])
            count += 1
        done.append(count)

    threads = [threading.Thread(target=worker, args=(k,)) for k in range(n_threads)]
    t0 = time.perf_counter()
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    seconds = time.perf_counter() - t0
    gb = sum(done) * q_count * 32 * ((n + 63) // 64 * 8) / 1e9
    print(f"{n_threads} host threads: {sum(done) / seconds:7.0f} launches/s, {sum(done) * q_count / seconds:9.0f} programs/s, {gb / seconds:6.0f} GB/s", flush=True)
