"""ctypes binding of include/silo_engine.h — the QueryEngine-shaped C++ host (lib/libsilo_engine.so).

Python here only marshals arguments; parsing, compilation of the filter tree, kernel launches and the
result post-processing all happen in the C++ host.  Raises if the native library is missing.
"""
import ctypes
import json
import os

import numpy as np

from . import binding

_LIB_PATH = os.path.join(binding._LIB_DIR, "libsilo_engine.so")

EXPORTED_SYMBOLS = [
    "silo_engine_create", "silo_engine_create_from_directory", "silo_engine_destroy", "silo_engine_add_partition", "silo_engine_append_sequences",
    "silo_engine_generate_synthetic", "silo_engine_build_pass", "silo_engine_set_lineage_column", "silo_engine_set_lineage_column_ids",
    "silo_engine_set_schema", "silo_engine_append_metadata", "silo_engine_append_unaligned_sequences", "silo_engine_finalize", "silo_engine_set_sharding", "silo_engine_set_comm", "silo_engine_set_broadcast", "silo_engine_set_option", "silo_engine_execute_query", "silo_engine_run_clients", "silo_engine_evaluate_filter", "silo_engine_execute_batch", "silo_engine_free_string", "silo_engine_data_version",
    "silo_engine_last_timings", "silo_engine_last_trace", "silo_engine_partition_store", "silo_engine_seqstore_id", "silo_engine_position_window",
    "silo_engine_last_error",
]

ALL_REDUCE_FN = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p)
BROADCAST_FN = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_uint32, ctypes.c_void_p)

_lib = None


def load_library():
    global _lib
    if _lib is not None:
        return _lib
    binding.load_library()  # libsilo_gpu.so first (same directory, $ORIGIN rpath)
    if not os.path.exists(_LIB_PATH):
        raise ImportError(f"{_LIB_PATH} is missing: run __graft_entry__.build() first; there is no CPU fallback")
    lib = ctypes.CDLL(_LIB_PATH)
    vp = ctypes.c_void_p
    lib.silo_engine_create.argtypes = [ctypes.c_char_p, ctypes.c_char_p, ctypes.c_char_p, ctypes.c_int, ctypes.POINTER(vp)]
    lib.silo_engine_create_from_directory.argtypes = [ctypes.c_char_p, ctypes.c_int, ctypes.POINTER(vp), ctypes.POINTER(vp)]
    lib.silo_engine_destroy.argtypes = [vp]
    lib.silo_engine_destroy.restype = None
    lib.silo_engine_add_partition.argtypes = [vp, ctypes.c_uint32]
    lib.silo_engine_append_sequences.argtypes = [vp, ctypes.c_int, ctypes.c_char_p, ctypes.c_int, ctypes.c_uint32, ctypes.c_uint32, vp, vp]
    lib.silo_engine_generate_synthetic.argtypes = [vp, ctypes.c_int, ctypes.c_char_p, ctypes.c_int, ctypes.POINTER(binding.SynthDesc)]
    lib.silo_engine_build_pass.argtypes = [vp, ctypes.c_int, ctypes.c_char_p, ctypes.c_int, ctypes.c_int]
    lib.silo_engine_set_lineage_column.argtypes = [vp, ctypes.c_int, ctypes.c_char_p, ctypes.POINTER(ctypes.c_char_p), ctypes.c_uint32]
    lib.silo_engine_set_lineage_column_ids.argtypes = [
        vp, ctypes.c_int, ctypes.c_char_p, ctypes.POINTER(ctypes.c_char_p), ctypes.c_uint32, vp, ctypes.c_uint32]
    lib.silo_engine_finalize.argtypes = [vp]
    lib.silo_engine_set_sharding.argtypes = [vp, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_int, ALL_REDUCE_FN, vp]
    lib.silo_engine_set_broadcast.argtypes = [vp, BROADCAST_FN, vp]
    lib.silo_engine_set_comm.argtypes = [vp, vp, ctypes.c_int]
    lib.silo_engine_evaluate_filter.argtypes = [vp, ctypes.c_char_p, ctypes.c_int, vp, ctypes.c_size_t, ctypes.POINTER(ctypes.c_uint32), ctypes.POINTER(vp),
                                                ctypes.POINTER(ctypes.c_int)]
    lib.silo_engine_set_schema.argtypes = [vp, ctypes.c_char_p, ctypes.c_char_p]
    lib.silo_engine_append_metadata.argtypes = [vp, ctypes.c_int, ctypes.c_char_p, ctypes.c_char_p, ctypes.POINTER(ctypes.c_char_p), ctypes.c_uint32]
    lib.silo_engine_append_unaligned_sequences.argtypes = [vp, ctypes.c_int, ctypes.c_char_p, ctypes.POINTER(ctypes.c_char_p), ctypes.c_uint32]
    lib.silo_engine_data_version.argtypes = [vp, ctypes.POINTER(vp)]
    lib.silo_engine_set_option.argtypes = [vp, ctypes.c_char_p, ctypes.c_int64]
    lib.silo_engine_execute_query.argtypes = [vp, ctypes.c_char_p, ctypes.POINTER(vp), ctypes.POINTER(ctypes.c_int)]
    lib.silo_engine_execute_batch.argtypes = [vp, ctypes.POINTER(ctypes.c_char_p), ctypes.c_uint32, ctypes.POINTER(vp), ctypes.POINTER(ctypes.c_int)]
    lib.silo_engine_run_clients.argtypes = [vp, ctypes.c_char_p, ctypes.c_uint32, ctypes.c_double, ctypes.POINTER(ctypes.c_uint64), ctypes.POINTER(ctypes.c_double),
                                            ctypes.POINTER(vp)]
    lib.silo_engine_free_string.argtypes = [vp]
    lib.silo_engine_free_string.restype = None
    lib.silo_engine_last_timings.argtypes = [ctypes.POINTER(ctypes.c_int64), ctypes.POINTER(ctypes.c_int64)]
    lib.silo_engine_last_timings.restype = None
    lib.silo_engine_last_trace.argtypes = [ctypes.POINTER(vp)]
    lib.silo_engine_partition_store.argtypes = [vp, ctypes.c_int]
    lib.silo_engine_partition_store.restype = vp
    lib.silo_engine_seqstore_id.argtypes = [vp, ctypes.c_int, ctypes.c_char_p, ctypes.c_int]
    lib.silo_engine_position_window.argtypes = [vp, ctypes.c_char_p, ctypes.c_int, ctypes.POINTER(ctypes.c_uint32), ctypes.POINTER(ctypes.c_uint32)]
    lib.silo_engine_last_error.restype = ctypes.c_char_p
    _lib = lib
    return lib


class SiloEngineError(RuntimeError):
    pass


class QueryError(RuntimeError):
    """Non-200 answer of execute_query: .status is 400 / 500, .document the error JSON."""

    def __init__(self, status, document):
        super().__init__(f"{status}: {document}")
        self.status = status
        self.document = document


def _check(rc):
    if rc < 0:
        raise SiloEngineError(load_library().silo_engine_last_error().decode())
    return rc


class StoreView:
    """Borrowed view of a partition's silo_gpu_store for direct kernel calls (bench roofline leg)."""

    def __init__(self, handle):
        self.lib = binding.load_library()
        self.handle = ctypes.c_void_p(handle)
        self.row_words = self.lib.silo_gpu_store_row_words(self.handle)
        self.sequence_count = self.lib.silo_gpu_store_sequence_count(self.handle)
        self.device_bytes = self.lib.silo_gpu_store_device_bytes(self.handle)


class Engine:
    def __init__(self, reference_genomes, alias=None, default_nucleotide_sequence=None, device=0):
        """reference_genomes / alias: dicts (or JSON strings) in the reference's file formats."""
        self.lib = load_library()
        genomes_text = reference_genomes if isinstance(reference_genomes, str) else json.dumps(reference_genomes)
        alias_text = None if alias is None else (alias if isinstance(alias, str) else json.dumps(alias))
        handle = ctypes.c_void_p()
        _check(self.lib.silo_engine_create(
            genomes_text.encode(), None if alias_text is None else alias_text.encode(),
            None if default_nucleotide_sequence is None else default_nucleotide_sequence.encode(), device, ctypes.byref(handle)))
        self.handle = handle
        self._callbacks = []

    @classmethod
    def from_directory(cls, directory, device=0):
        """Loads a data set directory in the reference's input formats (see include/silo_engine.h)."""
        self = cls.__new__(cls)
        self.lib = load_library()
        self._callbacks = []
        handle = ctypes.c_void_p()
        summary = ctypes.c_void_p()
        _check(self.lib.silo_engine_create_from_directory(str(directory).encode(), device, ctypes.byref(handle), ctypes.byref(summary)))
        self.handle = handle
        try:
            self.summary = json.loads(ctypes.string_at(summary).decode())
        finally:
            self.lib.silo_engine_free_string(summary)
        return self

    def close(self):
        if getattr(self, "handle", None):
            self.lib.silo_engine_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def add_partition(self, sequence_count):
        return _check(self.lib.silo_engine_add_partition(self.handle, sequence_count))

    def append_sequences(self, partition, name, is_aa, first_sequence, sequences):
        """sequences: list of str / None (missing genome)."""
        n = len(sequences)
        if n == 0:
            return
        length = next((len(s) for s in sequences if s is not None), 0)
        chars = np.zeros((n, max(length, 1)), dtype=np.uint8)
        is_null = np.zeros(n, dtype=np.uint8)
        for i, seq in enumerate(sequences):
            if seq is None:
                is_null[i] = 1
            else:
                if len(seq) != length:
                    raise ValueError("sequences of one store must have equal length")
                chars[i, :length] = np.frombuffer(seq.encode("latin-1"), dtype=np.uint8)
        _check(self.lib.silo_engine_append_sequences(
            self.handle, partition, name.encode(), int(is_aa), first_sequence, n, chars.ctypes.data_as(ctypes.c_void_p),
            is_null.ctypes.data_as(ctypes.c_void_p)))

    def generate_synthetic(self, partition, name, is_aa, model, window=None):
        """window = (begin, end): the position-range shard this rank's store holds (None = all)."""
        begin, end = window if window is not None else (0, model.positions)
        arrays = [
            np.ascontiguousarray(model.lineage_of_sequence, dtype=np.uint16), np.ascontiguousarray(model.lead_gap, dtype=np.uint32),
            np.ascontiguousarray(model.trail_gap, dtype=np.uint32), np.ascontiguousarray(model.missing_start, dtype=np.uint32),
            np.ascontiguousarray(model.missing_len, dtype=np.uint32),
            np.ascontiguousarray(model.lineage_symbol[begin:end], dtype=np.uint8),
        ]
        desc = binding.SynthDesc(
            model.seed, model.n_lineages, arrays[0].ctypes.data_as(binding.c_u16p), arrays[1].ctypes.data_as(binding.c_u32p),
            arrays[2].ctypes.data_as(binding.c_u32p), arrays[3].ctypes.data_as(binding.c_u32p), arrays[4].ctypes.data_as(binding.c_u32p),
            arrays[5].ctypes.data_as(binding.c_u8p), model.private_threshold, model.ambiguous_threshold, begin, model.positions)
        _check(self.lib.silo_engine_generate_synthetic(self.handle, partition, name.encode(), int(is_aa), ctypes.byref(desc)))

    def build_pass(self, partition, name, is_aa, which):
        """Two-pass build of a sequence store: 1 = the appends that follow are only counted, 2 = repeated, they are written
        straight into the adaptive planes (silo_engine_build_pass)."""
        _check(self.lib.silo_engine_build_pass(self.handle, partition, name.encode(), int(is_aa), which))

    def set_lineage_column(self, partition, column, values):
        array = (ctypes.c_char_p * len(values))(*[None if v is None else v.encode() for v in values])
        _check(self.lib.silo_engine_set_lineage_column(self.handle, partition, column.encode(), array, len(values)))

    def set_lineage_column_ids(self, partition, column, dictionary, value_ids):
        names = (ctypes.c_char_p * len(dictionary))(*[d.encode() for d in dictionary])
        ids = np.ascontiguousarray(value_ids, dtype=np.uint32)
        _check(self.lib.silo_engine_set_lineage_column_ids(
            self.handle, partition, column.encode(), names, len(dictionary), ids.ctypes.data_as(ctypes.c_void_p), len(ids)))

    def finalize(self):
        _check(self.lib.silo_engine_finalize(self.handle))

    def set_sharding(self, rank, world, shard_by_position, all_reduce=None):
        """all_reduce(device_ptr:int, n:int, stream) -> sums n uint32 in place across ranks."""
        if all_reduce is None:
            callback = ALL_REDUCE_FN(0)
        else:
            def trampoline(_context, device_values, n, stream):
                try:
                    all_reduce(device_values, n, stream)
                    return 0
                except Exception as error:  # never let an exception cross the C boundary
                    print("all_reduce callback failed:", error)
                    return 1
            callback = ALL_REDUCE_FN(trampoline)
        self._callbacks.append(callback)
        _check(self.lib.silo_engine_set_sharding(self.handle, rank, world, int(shard_by_position), callback, None))

    def set_comm(self, comm, shard_by_position):
        """Native collectives (binding.Comm = RCCL over xGMI) on the engine's own streams; rank / world are the communicator's."""
        self._callbacks.append(comm)  # the communicator must outlive the engine
        _check(self.lib.silo_engine_set_comm(self.handle, comm.handle, int(shard_by_position)))

    def set_broadcast(self, broadcast):
        """broadcast(device_ptr:int, nbytes:int, root:int, stream): in-place broadcast from rank `root`."""
        def trampoline(_context, device_bytes, nbytes, root, stream):
            try:
                broadcast(device_bytes, nbytes, root, stream)
                return 0
            except Exception as error:  # never let an exception cross the C boundary
                print("broadcast callback failed:", error)
                return 1
        callback = BROADCAST_FN(trampoline)
        self._callbacks.append(callback)
        _check(self.lib.silo_engine_set_broadcast(self.handle, callback, None))

    def set_schema(self, primary_key, date_to_sort_by=None):
        _check(self.lib.silo_engine_set_schema(self.handle, primary_key.encode(), date_to_sort_by.encode() if date_to_sort_by else None))

    def append_metadata(self, partition, column, column_type, values):
        """values: texts as the metadata TSV holds them (None / '' = null); column_type as in database_config.yaml
        plus 'indexed_string' for a string column with generateIndex."""
        array = (ctypes.c_char_p * max(len(values), 1))(*[None if v is None else str(v).encode() for v in values])
        _check(self.lib.silo_engine_append_metadata(self.handle, partition, column.encode(), column_type.encode(), array, len(values)))

    def append_unaligned_sequences(self, partition, sequence_name, sequences):
        """Unaligned nucleotide sequences (None = none) for the Fasta action."""
        array = (ctypes.c_char_p * max(len(sequences), 1))(*[None if s is None else s.encode() for s in sequences])
        _check(self.lib.silo_engine_append_unaligned_sequences(self.handle, partition, sequence_name.encode(), array, len(sequences)))

    def data_version(self):
        out = ctypes.c_void_p()
        _check(self.lib.silo_engine_data_version(self.handle, ctypes.byref(out)))
        try:
            return ctypes.string_at(out).decode()
        finally:
            self.lib.silo_engine_free_string(out)

    def set_option(self, name, value):
        _check(self.lib.silo_engine_set_option(self.handle, name.encode(), int(value)))

    def execute_text(self, query):
        """Returns (http_status, response body as bytes) — what silo_api would put on the wire."""
        text = query if isinstance(query, (str, bytes)) else json.dumps(query)
        if isinstance(text, str):
            text = text.encode()
        out = ctypes.c_void_p()
        status = ctypes.c_int()
        _check(self.lib.silo_engine_execute_query(self.handle, text, ctypes.byref(out), ctypes.byref(status)))
        try:
            body = ctypes.string_at(out)
        finally:
            self.lib.silo_engine_free_string(out)
        return status.value, body

    def run_clients(self, query, n_clients, seconds):
        """silo_engine_run_clients: native request threads, one query at a time each -> (queries per second, last response body)."""
        text = query if isinstance(query, (str, bytes)) else json.dumps(query)
        if isinstance(text, str):
            text = text.encode()
        answered, elapsed, out = ctypes.c_uint64(), ctypes.c_double(), ctypes.c_void_p()
        _check(self.lib.silo_engine_run_clients(self.handle, text, n_clients, float(seconds), ctypes.byref(answered), ctypes.byref(elapsed), ctypes.byref(out)))
        try:
            body = ctypes.string_at(out)
        finally:
            self.lib.silo_engine_free_string(out)
        return answered.value / elapsed.value, body

    def evaluate_filter(self, expression, partition=0, n_rows=None):
        """Operator::evaluate for one partition: (bitset as uint64 words, cardinality).  n_rows = the partition's row count."""
        text = expression if isinstance(expression, (str, bytes)) else json.dumps(expression)
        if isinstance(text, str):
            text = text.encode()
        if n_rows is None:
            n_rows = self.partition_store(partition).sequence_count
        words = np.zeros((n_rows + 63) // 64, dtype=np.uint64)
        count, status, error = ctypes.c_uint32(), ctypes.c_int(), ctypes.c_void_p()
        _check(self.lib.silo_engine_evaluate_filter(
            self.handle, text, partition, words.ctypes.data_as(ctypes.c_void_p), len(words), ctypes.byref(count), ctypes.byref(error), ctypes.byref(status)))
        if status.value != 200:
            try:
                document = json.loads(ctypes.string_at(error).decode())
            finally:
                self.lib.silo_engine_free_string(error)
            raise QueryError(status.value, document)
        return words, count.value

    def execute_batch_text(self, queries):
        """One silo_engine_execute_batch call: [(http_status, response body as bytes)] in the order of `queries`."""
        texts = [q if isinstance(q, (str, bytes)) else json.dumps(q) for q in queries]
        texts = [t.encode() if isinstance(t, str) else t for t in texts]
        n = len(texts)
        array = (ctypes.c_char_p * max(n, 1))(*texts)
        outs = (ctypes.c_void_p * max(n, 1))()
        statuses = (ctypes.c_int * max(n, 1))()
        _check(self.lib.silo_engine_execute_batch(self.handle, array, n, outs, statuses))
        results = []
        for i in range(n):
            try:
                results.append((statuses[i], ctypes.string_at(outs[i])))
            finally:
                self.lib.silo_engine_free_string(outs[i])
        return results

    def execute_batch(self, queries):
        """[(http_status, parsed JSON document)] of a batch of queries sharing passes over the planes."""
        return [(status, json.loads(body.decode())) for status, body in self.execute_batch_text(queries)]

    def execute_raw(self, query):
        """Returns (http_status, parsed JSON document)."""
        status, body = self.execute_text(query)
        return status, json.loads(body.decode())

    def execute_query(self, query):
        status, document = self.execute_raw(query)
        if status != 200:
            raise QueryError(status, document)
        return document["queryResult"]

    def last_timings(self):
        filter_us, action_us = ctypes.c_int64(), ctypes.c_int64()
        self.lib.silo_engine_last_timings(ctypes.byref(filter_us), ctypes.byref(action_us))
        return filter_us.value, action_us.value

    def last_trace(self):
        """Phase marks (µs since the query began) of the last query on this thread."""
        out = ctypes.c_void_p()
        _check(self.lib.silo_engine_last_trace(ctypes.byref(out)))
        try:
            return json.loads(ctypes.string_at(out).decode())
        finally:
            self.lib.silo_engine_free_string(out)

    def position_window(self, name, is_aa):
        begin, end = ctypes.c_uint32(), ctypes.c_uint32()
        _check(self.lib.silo_engine_position_window(self.handle, name.encode(), int(is_aa), ctypes.byref(begin), ctypes.byref(end)))
        return begin.value, end.value

    def partition_store(self, partition):
        return StoreView(self.lib.silo_engine_partition_store(self.handle, partition))

    def seqstore_id(self, partition, name, is_aa):
        return self.lib.silo_engine_seqstore_id(self.handle, partition, name.encode(), int(is_aa))
