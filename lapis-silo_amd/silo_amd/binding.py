"""ctypes binding of include/silo_gpu.h (the C-ABI drop-in boundary).

This is the stub a maintainer of a Python host would write; the reference's own host is C++ and binds
the same symbols directly.  There is no CPU fallback: if the HIP library is missing or
no GPU is visible every entry point raises.
"""
import ctypes
import os

import numpy as np

from . import alphabet

_LIB_DIR = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "lib")
_LIB_PATH = os.path.join(_LIB_DIR, "libsilo_gpu.so")

c_u8p = ctypes.POINTER(ctypes.c_uint8)
c_u16p = ctypes.POINTER(ctypes.c_uint16)
c_u32p = ctypes.POINTER(ctypes.c_uint32)
c_u64p = ctypes.POINTER(ctypes.c_uint64)


class SiloGpuError(RuntimeError):
    def __init__(self, code, message):
        super().__init__(f"silo_gpu error {code}: {message}")
        self.code = code


class SeqStoreDesc(ctypes.Structure):
    _fields_ = [
        ("alphabet", ctypes.c_uint32),
        ("positions", ctypes.c_uint32),
        ("reference", c_u8p),
        ("n_scan_symbols", ctypes.c_uint32),
        ("scan_symbols", c_u8p),
        ("n_extra_symbols", ctypes.c_uint32),
        ("extra_symbols", c_u8p),
    ]


class StoreDesc(ctypes.Structure):
    _fields_ = [
        ("device", ctypes.c_int32),
        ("sequence_count", ctypes.c_uint32),
        ("n_seqstores", ctypes.c_uint32),
        ("seqstores", ctypes.POINTER(SeqStoreDesc)),
    ]


class SynthDesc(ctypes.Structure):
    _fields_ = [
        ("seed", ctypes.c_uint64),
        ("n_lineages", ctypes.c_uint32),
        ("lineage_of_sequence", c_u16p),
        ("lead_gap", c_u32p),
        ("trail_gap", c_u32p),
        ("missing_start", c_u32p),
        ("missing_len", c_u32p),
        ("lineage_symbol", c_u8p),
        ("private_threshold", ctypes.c_uint32),
        ("ambiguous_threshold", ctypes.c_uint32),
        ("position_offset", ctypes.c_uint32),
        ("total_positions", ctypes.c_uint32),
    ]


class RoaringPayload(ctypes.Structure):
    _fields_ = [("symbol", ctypes.c_uint32), ("bytes", ctypes.c_char_p), ("n_bytes", ctypes.c_size_t)]


class ScanTiming(ctypes.Structure):
    """silo_gpu_scan_timing: one plane-scan launch of the thread's last Mutations scan."""
    _fields_ = [("kernel", ctypes.c_char * 64), ("plane_rows", ctypes.c_uint64), ("bytes", ctypes.c_uint64), ("filters", ctypes.c_uint32),
                ("blocks", ctypes.c_uint32), ("ms", ctypes.c_float)]


def scan_timings(capacity=64):
    """Per-launch timings of this thread's last scan (SILO_GPU_TUNE_SCAN_TIMING = knob 7 set to 1 before it)."""
    lib = load_library()
    out = (ScanTiming * capacity)()
    n = ctypes.c_uint32()
    _check(lib.silo_gpu_scan_timings(out, capacity, ctypes.byref(n)))
    return [dict(kernel=out[k].kernel.decode(), plane_rows=int(out[k].plane_rows), bytes=int(out[k].bytes), filters=int(out[k].filters), blocks=int(out[k].blocks),
                 ms=float(out[k].ms))
            for k in range(min(capacity, n.value))]


class BitProg(ctypes.Structure):
    _fields_ = [
        ("n_instructions", ctypes.c_uint32),
        ("code", c_u32p),
        ("n_leaves", ctypes.c_uint32),
        ("leaves", ctypes.POINTER(ctypes.c_void_p)),
        ("n_slots", ctypes.c_uint32),
    ]


# every symbol include/silo_gpu.h declares; tests check the library exports all of them
EXPORTED_SYMBOLS = [
    "silo_gpu_store_create", "silo_gpu_store_destroy", "silo_gpu_store_set_options", "silo_gpu_store_sequence_count",
    "silo_gpu_store_row_words", "silo_gpu_store_device_bytes", "silo_gpu_store_append_sequences",
    "silo_gpu_store_finalize", "silo_gpu_store_generate_synthetic", "silo_gpu_bitset_alloc",
    "silo_gpu_bitset_upload", "silo_gpu_bitset_download", "silo_gpu_bitset_from_lineages", "silo_gpu_upload_u32",
    "silo_gpu_bitset_from_value_ids", "silo_gpu_free",
    "silo_gpu_malloc", "silo_gpu_memcpy_d2h", "silo_gpu_memcpy_h2d", "silo_gpu_stream_synchronize", "silo_gpu_stream_create", "silo_gpu_set_device", "silo_gpu_stream_destroy", "silo_gpu_store_plane",
    "silo_gpu_store_sparse_plane", "silo_gpu_filter_eval", "silo_gpu_filter_eval_batch", "silo_gpu_popcount", "silo_gpu_mutations_scan", "silo_gpu_mutations_scan_batch", "silo_gpu_mutations_scan_ranges", "silo_gpu_store_scan_planes", "silo_gpu_store_scan_escapes", "silo_gpu_store_scan_rows", "silo_gpu_store_scan_runs", "silo_gpu_row_slot_create", "silo_gpu_row_slot_destroy", "silo_gpu_mutations_select_to_slot", "silo_gpu_row_slot_wait", "silo_gpu_store_scan_sparse_keys", "silo_gpu_store_finalize_seqstore", "silo_gpu_store_build_pass", "silo_gpu_store_build_mode", "silo_gpu_store_memory_info", "silo_gpu_store_import_position", "silo_gpu_store_import_missing_rows",
    "silo_gpu_memset_async", "silo_gpu_event_create", "silo_gpu_event_record", "silo_gpu_event_elapsed_ms",
    "silo_gpu_event_destroy", "silo_gpu_event_synchronize", "silo_gpu_host_alloc", "silo_gpu_host_free", "silo_gpu_memcpy_d2h_async", "silo_gpu_mutations_select", "silo_gpu_upload_bytes", "silo_gpu_upload_column", "silo_gpu_bitset_from_compare", "silo_gpu_group_count", "silo_gpu_group_count_hashed", "silo_gpu_reconstruct_sequences", "silo_gpu_bitset_from_pairs", "silo_gpu_count_pairs", "silo_gpu_count_slot_create", "silo_gpu_count_slot_destroy", "silo_gpu_filter_eval_count", "silo_gpu_count_slot_wait", "silo_gpu_tune", "silo_gpu_last_scan_kernel", "silo_gpu_scan_timings", "silo_gpu_stream_read_probe", "silo_gpu_last_error",
    "silo_gpu_comm_unique_id", "silo_gpu_comm_create", "silo_gpu_comm_destroy", "silo_gpu_comm_rank", "silo_gpu_comm_world",
    "silo_gpu_allreduce_counts", "silo_gpu_broadcast_bytes",
]

_lib = None


def load_library():
    """Loads lib/libsilo_gpu.so; raises loudly if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(_LIB_PATH):
        raise ImportError(
            f"{_LIB_PATH} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950). The product path has no CPU fallback."
        )
    lib = ctypes.CDLL(_LIB_PATH)
    vp = ctypes.c_void_p
    lib.silo_gpu_store_create.argtypes = [ctypes.POINTER(StoreDesc), ctypes.POINTER(vp)]
    lib.silo_gpu_store_destroy.argtypes = [vp]
    lib.silo_gpu_store_destroy.restype = None
    lib.silo_gpu_store_sequence_count.argtypes = [vp]
    lib.silo_gpu_store_sequence_count.restype = ctypes.c_uint32
    lib.silo_gpu_store_row_words.argtypes = [vp]
    lib.silo_gpu_store_row_words.restype = ctypes.c_uint32
    lib.silo_gpu_store_device_bytes.argtypes = [vp]
    lib.silo_gpu_store_device_bytes.restype = ctypes.c_uint64
    lib.silo_gpu_store_append_sequences.argtypes = [vp, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_uint32, vp, vp]
    lib.silo_gpu_store_finalize.argtypes = [vp]
    lib.silo_gpu_store_generate_synthetic.argtypes = [vp, ctypes.c_uint32, ctypes.POINTER(SynthDesc)]
    lib.silo_gpu_bitset_alloc.argtypes = [vp, ctypes.POINTER(vp)]
    lib.silo_gpu_bitset_upload.argtypes = [vp, vp, vp, ctypes.c_size_t, vp]
    lib.silo_gpu_bitset_download.argtypes = [vp, vp, vp, ctypes.c_size_t, vp]
    lib.silo_gpu_bitset_from_lineages.argtypes = [vp, vp, vp, ctypes.c_uint32, vp]
    lib.silo_gpu_free.argtypes = [vp]
    lib.silo_gpu_free.restype = None
    lib.silo_gpu_malloc.argtypes = [ctypes.c_size_t, ctypes.POINTER(vp)]
    lib.silo_gpu_memcpy_d2h.argtypes = [vp, vp, ctypes.c_size_t, vp]
    lib.silo_gpu_memcpy_h2d.argtypes = [vp, vp, ctypes.c_size_t, vp]
    lib.silo_gpu_upload_u32.argtypes = [vp, ctypes.c_size_t, ctypes.POINTER(vp)]
    lib.silo_gpu_bitset_from_value_ids.argtypes = [vp, vp, vp, vp, ctypes.c_uint32, vp]
    lib.silo_gpu_stream_synchronize.argtypes = [vp]
    lib.silo_gpu_stream_create.argtypes = [ctypes.POINTER(vp)]
    lib.silo_gpu_stream_destroy.argtypes = [vp]
    lib.silo_gpu_stream_destroy.restype = None
    lib.silo_gpu_store_plane.argtypes = [vp, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_uint32]
    lib.silo_gpu_store_plane.restype = vp
    lib.silo_gpu_store_sparse_plane.argtypes = [vp, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_uint32, vp, vp]
    lib.silo_gpu_filter_eval.argtypes = [vp, ctypes.POINTER(BitProg), vp, vp, vp]
    lib.silo_gpu_popcount.argtypes = [vp, vp, vp, vp]
    lib.silo_gpu_mutations_scan.argtypes = [vp, ctypes.c_uint32, vp, ctypes.c_uint32, ctypes.c_uint32, vp, vp]
    lib.silo_gpu_mutations_scan_batch.argtypes = [vp, ctypes.c_uint32, ctypes.POINTER(vp), ctypes.c_uint32, ctypes.c_uint32, ctypes.c_uint32, ctypes.POINTER(vp), vp]
    lib.silo_gpu_store_scan_planes.argtypes = [vp, ctypes.c_uint32]
    lib.silo_gpu_store_scan_planes.restype = ctypes.c_uint32
    lib.silo_gpu_store_scan_escapes.argtypes = [vp, ctypes.c_uint32]
    lib.silo_gpu_store_scan_escapes.restype = ctypes.c_uint64
    lib.silo_gpu_mutations_scan_ranges.argtypes = [vp, ctypes.POINTER(ctypes.c_uint32), ctypes.c_uint32, ctypes.POINTER(vp), ctypes.c_uint32, ctypes.POINTER(vp), vp]
    lib.silo_gpu_memset_async.argtypes = [vp, ctypes.c_int, ctypes.c_size_t, vp]
    lib.silo_gpu_upload_column.argtypes = [vp, ctypes.c_size_t, ctypes.c_int, ctypes.POINTER(vp)]
    lib.silo_gpu_bitset_from_compare.argtypes = [vp, vp, vp, ctypes.c_int, ctypes.c_int, vp, vp]
    lib.silo_gpu_group_count.argtypes = [vp, vp, ctypes.POINTER(vp), ctypes.POINTER(ctypes.c_uint32), ctypes.c_uint32, vp, vp]
    lib.silo_gpu_mutations_select.argtypes = [vp, vp, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_double, ctypes.c_uint32, vp, vp]
    lib.silo_gpu_upload_bytes.argtypes = [vp, ctypes.c_size_t, ctypes.POINTER(vp)]
    lib.silo_gpu_group_count_hashed.argtypes = [vp, vp, ctypes.POINTER(vp), ctypes.POINTER(ctypes.c_uint32), ctypes.c_uint32, ctypes.c_uint32,
                                                ctypes.POINTER(vp), ctypes.POINTER(vp), ctypes.POINTER(ctypes.c_uint32), vp]
    lib.silo_gpu_reconstruct_sequences.argtypes = [vp, ctypes.c_uint32, vp, ctypes.c_uint32, vp, vp]
    lib.silo_gpu_event_create.argtypes = [ctypes.POINTER(vp)]
    lib.silo_gpu_event_record.argtypes = [vp, vp]
    lib.silo_gpu_event_elapsed_ms.argtypes = [vp, vp, ctypes.POINTER(ctypes.c_float)]
    lib.silo_gpu_event_destroy.argtypes = [vp]
    lib.silo_gpu_event_destroy.restype = None
    lib.silo_gpu_tune.argtypes = [ctypes.c_int, ctypes.c_int]
    lib.silo_gpu_last_scan_kernel.restype = ctypes.c_char_p
    lib.silo_gpu_store_build_pass.argtypes = [vp, ctypes.c_uint32, ctypes.c_int]
    lib.silo_gpu_store_build_pass.restype = ctypes.c_int
    lib.silo_gpu_store_build_mode.argtypes = [vp, ctypes.c_uint32]
    lib.silo_gpu_store_build_mode.restype = ctypes.c_int
    lib.silo_gpu_scan_timings.argtypes = [ctypes.POINTER(ScanTiming), ctypes.c_uint32, ctypes.POINTER(ctypes.c_uint32)]
    lib.silo_gpu_scan_timings.restype = ctypes.c_int
    lib.silo_gpu_last_error.restype = ctypes.c_char_p
    lib.silo_gpu_stream_read_probe.argtypes = [ctypes.c_uint64, ctypes.c_uint32, ctypes.POINTER(ctypes.c_float)]
    lib.silo_gpu_stream_read_probe.restype = ctypes.c_int
    lib.silo_gpu_store_scan_rows.argtypes = [vp, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_uint32]
    lib.silo_gpu_store_scan_rows.restype = ctypes.c_uint64
    for name in ("silo_gpu_store_scan_runs", "silo_gpu_store_scan_sparse_keys"):
        getattr(lib, name).argtypes = [vp, ctypes.c_uint32]
        getattr(lib, name).restype = ctypes.c_uint64
    lib.silo_gpu_store_finalize_seqstore.argtypes = [vp, ctypes.c_uint32]
    lib.silo_gpu_store_import_position.argtypes = [vp, ctypes.c_uint32, ctypes.c_uint32, ctypes.POINTER(RoaringPayload), ctypes.c_uint32, ctypes.c_uint32,
                                                   ctypes.c_uint32]
    lib.silo_gpu_store_import_missing_rows.argtypes = [vp, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_uint32, ctypes.POINTER(RoaringPayload)]
    lib.silo_gpu_filter_eval_batch.argtypes = [vp, ctypes.POINTER(BitProg), ctypes.c_uint32, ctypes.POINTER(vp), c_u64p, vp]
    lib.silo_gpu_comm_unique_id.argtypes = [vp]
    lib.silo_gpu_comm_create.argtypes = [vp, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_int, ctypes.POINTER(vp)]
    lib.silo_gpu_comm_destroy.argtypes = [vp]
    lib.silo_gpu_comm_destroy.restype = None
    lib.silo_gpu_comm_rank.argtypes = [vp]
    lib.silo_gpu_comm_rank.restype = ctypes.c_uint32
    lib.silo_gpu_comm_world.argtypes = [vp]
    lib.silo_gpu_comm_world.restype = ctypes.c_uint32
    lib.silo_gpu_allreduce_counts.argtypes = [vp, vp, ctypes.c_size_t, vp]
    lib.silo_gpu_broadcast_bytes.argtypes = [vp, vp, ctypes.c_size_t, ctypes.c_uint32, vp]
    _lib = lib
    return lib


COMM_ID_BYTES = 128


def comm_unique_id():
    """The opaque id one rank creates and hands to the others out of band (silo_gpu_comm_unique_id)."""
    buffer = (ctypes.c_uint8 * COMM_ID_BYTES)()
    _check(load_library().silo_gpu_comm_unique_id(buffer))
    return bytes(buffer)


class Comm:
    """silo_gpu_comm: the native RCCL communicator behind silo_gpu_allreduce_counts / silo_gpu_broadcast_bytes."""

    def __init__(self, unique_id, rank, world, device=0):
        self.lib = load_library()
        if len(unique_id) != COMM_ID_BYTES:
            raise ValueError("a communicator id has %d bytes" % COMM_ID_BYTES)
        handle = ctypes.c_void_p()
        _check(self.lib.silo_gpu_comm_create((ctypes.c_uint8 * COMM_ID_BYTES).from_buffer_copy(unique_id), rank, world, device, ctypes.byref(handle)))
        self.handle = handle
        self.rank, self.world = rank, world

    def all_reduce_counts(self, device_ptr, n, stream=None):
        _check(self.lib.silo_gpu_allreduce_counts(self.handle, device_ptr, n, stream))

    def broadcast_bytes(self, device_ptr, nbytes, root, stream=None):
        _check(self.lib.silo_gpu_broadcast_bytes(self.handle, device_ptr, nbytes, root, stream))

    def close(self):
        if getattr(self, "handle", None):
            self.lib.silo_gpu_comm_destroy(self.handle)
            self.handle = None


def _check(rc):
    if rc != 0:
        raise SiloGpuError(rc, load_library().silo_gpu_last_error().decode())


def _ptr(array):
    return array.ctypes.data_as(ctypes.c_void_p)


class GpuEvent:
    """hipEvent on a caller-chosen stream (None = the null stream)."""

    def __init__(self):
        self.lib = load_library()
        self.handle = ctypes.c_void_p()
        _check(self.lib.silo_gpu_event_create(ctypes.byref(self.handle)))

    def record(self, stream=None):
        _check(self.lib.silo_gpu_event_record(self.handle, stream))

    def elapsed_ms(self, stop):
        ms = ctypes.c_float()
        _check(self.lib.silo_gpu_event_elapsed_ms(self.handle, stop.handle, ctypes.byref(ms)))
        return ms.value

    def __del__(self):
        if getattr(self, "handle", None):
            self.lib.silo_gpu_event_destroy(self.handle)
            self.handle = None


# bit-program opcodes (include/silo_gpu.h)
(OP_LOAD, OP_ZERO, OP_ONES, OP_NOT, OP_AND, OP_OR, OP_ANDNOT, OP_CNT_ADD, OP_CNT_GE, OP_CNT_EQ, OP_MOV,
 OP_OR_N, OP_AND_N, OP_CNT_ADD_N, OP_CNT_ADD_NOT_N) = range(15)


COUNT_SHARDS = 64   # SILO_GPU_COUNT_SHARDS
LEAF_OPERAND = 32   # SILO_GPU_LEAF_OPERAND


def encode(op, dst=0, a=0, b=0, imm=0):
    return [op | (dst << 8) | (a << 16) | (b << 24), imm]


class PreparedPrograms:
    """A batch of bit-programs marshalled once (ctypes arrays), so that repeated launches cost no Python work."""

    def __init__(self, programs, out_bitsets=None):
        self.n = len(programs)
        self.array = (BitProg * max(1, self.n))()
        self._keep = []
        for k, (code, leaves, n_slots) in enumerate(programs):
            code = np.ascontiguousarray(code, dtype=np.uint32)
            leaf_array = (ctypes.c_void_p * max(1, len(leaves)))(*[(l.value if isinstance(l, ctypes.c_void_p) else l) for l in leaves])
            self._keep += [code, leaf_array]
            self.array[k] = BitProg(len(code) // 2, code.ctypes.data_as(c_u32p), len(leaves), leaf_array, n_slots)
        self.outs = None
        if out_bitsets is not None:
            self.outs = (ctypes.c_void_p * max(1, self.n))(*[(o.value if isinstance(o, ctypes.c_void_p) else o) for o in out_bitsets])
        self.counts = np.zeros(max(1, self.n), dtype=np.uint64)

    def launch(self, store_handle, stream=None):
        _check(load_library().silo_gpu_filter_eval_batch(store_handle, self.array, self.n, self.outs, self.counts.ctypes.data_as(c_u64p), stream))
        return [int(c) for c in self.counts[:self.n]]


SYMBOL_NONE = 0xFF


def import_position(store_handle, seqstore_id, position, payloads, flipped=None, deleted=None):
    """silo_gpu_store_import_position: payloads = {symbol id: portable-format roaring bytes} of one reference Position."""
    items = sorted(payloads.items())
    array = (RoaringPayload * max(1, len(items)))(*[RoaringPayload(symbol, data, len(data)) for symbol, data in items])
    _check(load_library().silo_gpu_store_import_position(
        store_handle, seqstore_id, position, array, len(items), SYMBOL_NONE if flipped is None else flipped, SYMBOL_NONE if deleted is None else deleted))


def import_missing_rows(store_handle, seqstore_id, first_sequence, payloads):
    """silo_gpu_store_import_missing_rows: payloads[r] = portable-format roaring bytes of the POSITIONS where row first_sequence + r is missing."""
    array = (RoaringPayload * max(1, len(payloads)))(*[RoaringPayload(0, data, len(data)) for data in payloads])
    _check(load_library().silo_gpu_store_import_missing_rows(store_handle, seqstore_id, first_sequence, len(payloads), array))


def filter_eval_batch(store_handle, programs, out_bitsets=None, stream=None):
    """silo_gpu_filter_eval_batch on a raw store handle.  programs: list of (code, leaves, n_slots) with code a flat list of
    uint32 (2 per instruction) and leaves device pointers; returns the cardinalities (one launch for all of them)."""
    return PreparedPrograms(programs, out_bitsets).launch(store_handle, stream)


class GpuStore:
    """One device shard: the dense restatement of a DatabasePartition's sequence stores."""

    def __init__(self, sequence_count, seqstores, device=0):
        """seqstores: list of dicts {name, alphabet ('nuc'|'aa'), reference (np.uint8 symbol ids)}."""
        self.lib = load_library()
        self.sequence_count = int(sequence_count)
        self.names = [s["name"] for s in seqstores]
        self.alphabets = [s["alphabet"] for s in seqstores]
        self.references = [np.ascontiguousarray(s["reference"], dtype=np.uint8) for s in seqstores]
        self._keep = []
        descs = (SeqStoreDesc * len(seqstores))()
        self.scan_symbols = []
        for k, s in enumerate(seqstores):
            alpha = alphabet.ALPHABETS[s["alphabet"]]
            scan = np.array(s.get("scan_symbols", alpha.valid_mutation_symbols), dtype=np.uint8)
            extra = np.array(s.get("extra_symbols", [alpha.missing]), dtype=np.uint8)
            self._keep += [scan, extra]
            self.scan_symbols.append(scan)
            descs[k].alphabet = alpha.abi_id
            descs[k].positions = len(self.references[k])
            descs[k].reference = self.references[k].ctypes.data_as(c_u8p)
            descs[k].n_scan_symbols = len(scan)
            descs[k].scan_symbols = scan.ctypes.data_as(c_u8p)
            descs[k].n_extra_symbols = len(extra)
            descs[k].extra_symbols = extra.ctypes.data_as(c_u8p)
        desc = StoreDesc(device, self.sequence_count, len(seqstores), descs)
        handle = ctypes.c_void_p()
        _check(self.lib.silo_gpu_store_create(ctypes.byref(desc), ctypes.byref(handle)))
        self.handle = handle
        self.row_words = self.lib.silo_gpu_store_row_words(handle)
        self._owned = []

    # -- lifetime -------------------------------------------------------------------------------
    def close(self):
        if getattr(self, "handle", None):
            for ptr in self._owned:
                self.lib.silo_gpu_free(ptr)
            self._owned = []
            self.lib.silo_gpu_store_destroy(self.handle)
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

    @property
    def device_bytes(self):
        return self.lib.silo_gpu_store_device_bytes(self.handle)

    def seqstore_id(self, name):
        return self.names.index(name)

    def positions(self, seqstore_id):
        return len(self.references[seqstore_id])

    # -- build ----------------------------------------------------------------------------------
    def append_sequences(self, seqstore_id, first_sequence, sequences, is_null=None):
        """sequences: list of str/bytes (None = missing genome) or uint8 array [n][P] of characters."""
        positions = self.positions(seqstore_id)
        if isinstance(sequences, np.ndarray):
            chars = np.ascontiguousarray(sequences, dtype=np.uint8)
            n = chars.shape[0]
        else:
            n = len(sequences)
            chars = np.zeros((n, positions), dtype=np.uint8)
            if is_null is None:
                is_null = np.zeros(n, dtype=np.uint8)
            for i, seq in enumerate(sequences):
                if seq is None:
                    is_null[i] = 1
                    continue
                raw = seq.encode() if isinstance(seq, str) else bytes(seq)
                if len(raw) != positions:
                    raise ValueError(f"sequence {i} has length {len(raw)}, expected {positions}")
                chars[i] = np.frombuffer(raw, dtype=np.uint8)
        assert chars.shape == (n, positions)
        null_ptr = None
        if is_null is not None:
            is_null = np.ascontiguousarray(is_null, dtype=np.uint8)
            null_ptr = _ptr(is_null)
        _check(self.lib.silo_gpu_store_append_sequences(self.handle, seqstore_id, first_sequence, n, _ptr(chars), null_ptr))

    def generate_synthetic(self, seqstore_id, model):
        """model: silo_amd.synth.SynthModel for this sequence store."""
        arrays = dict(
            lineage=np.ascontiguousarray(model.lineage_of_sequence, dtype=np.uint16),
            lead=np.ascontiguousarray(model.lead_gap, dtype=np.uint32),
            trail=np.ascontiguousarray(model.trail_gap, dtype=np.uint32),
            mstart=np.ascontiguousarray(model.missing_start, dtype=np.uint32),
            mlen=np.ascontiguousarray(model.missing_len, dtype=np.uint32),
            table=np.ascontiguousarray(model.lineage_symbol, dtype=np.uint8),
        )
        assert len(arrays["lineage"]) == self.sequence_count
        assert arrays["table"].shape == (self.positions(seqstore_id), model.n_lineages)
        desc = SynthDesc(
            model.seed, model.n_lineages,
            arrays["lineage"].ctypes.data_as(c_u16p), arrays["lead"].ctypes.data_as(c_u32p),
            arrays["trail"].ctypes.data_as(c_u32p), arrays["mstart"].ctypes.data_as(c_u32p),
            arrays["mlen"].ctypes.data_as(c_u32p), arrays["table"].ctypes.data_as(c_u8p),
            model.private_threshold, model.ambiguous_threshold, 0, 0,
        )
        _check(self.lib.silo_gpu_store_generate_synthetic(self.handle, seqstore_id, ctypes.byref(desc)))

    def finalize(self):
        _check(self.lib.silo_gpu_store_finalize(self.handle))

    # -- device buffers -------------------------------------------------------------------------
    def bitset_alloc(self):
        ptr = ctypes.c_void_p()
        _check(self.lib.silo_gpu_bitset_alloc(self.handle, ctypes.byref(ptr)))
        self._owned.append(ptr)
        return ptr

    def malloc(self, nbytes):
        ptr = ctypes.c_void_p()
        _check(self.lib.silo_gpu_malloc(nbytes, ctypes.byref(ptr)))
        self._owned.append(ptr)
        return ptr

    def free(self, ptr):
        self._owned = [p for p in self._owned if p.value != ptr.value]
        self.lib.silo_gpu_free(ptr)

    def bitset_upload(self, ptr, words, stream=None):
        words = np.ascontiguousarray(words, dtype=np.uint64)
        _check(self.lib.silo_gpu_bitset_upload(self.handle, ptr, _ptr(words), len(words), stream))

    def bitset_download(self, ptr, stream=None):
        out = np.empty(self.row_words, dtype=np.uint64)
        _check(self.lib.silo_gpu_bitset_download(self.handle, _ptr(out), ptr, self.row_words, stream))
        return out

    def bitset_from_lineages(self, ptr, membership, stream=None):
        membership = np.ascontiguousarray(membership, dtype=np.uint8)
        _check(self.lib.silo_gpu_bitset_from_lineages(self.handle, ptr, _ptr(membership), len(membership), stream))

    def plane(self, seqstore_id, position, symbol):
        """Device pointer (int) of a dense plane or None for a sparse symbol."""
        return self.lib.silo_gpu_store_plane(self.handle, seqstore_id, position, symbol)

    def plane_download(self, seqstore_id, position, symbol):
        ptr = self.plane(seqstore_id, position, symbol)
        if ptr is None:
            tmp = self.bitset_alloc()
            _check(self.lib.silo_gpu_store_sparse_plane(self.handle, seqstore_id, position, symbol, tmp, None))
            out = self.bitset_download(tmp)
            self.free(tmp)
            return out
        return self.bitset_download(ctypes.c_void_p(ptr))

    def sparse_plane(self, seqstore_id, position, symbol, dst, stream=None):
        _check(self.lib.silo_gpu_store_sparse_plane(self.handle, seqstore_id, position, symbol, dst, stream))

    def memset(self, ptr, value, nbytes, stream=None):
        _check(self.lib.silo_gpu_memset_async(ptr, value, nbytes, stream))

    def read(self, ptr, dtype, count, stream=None):
        out = np.empty(count, dtype=dtype)
        _check(self.lib.silo_gpu_memcpy_d2h(_ptr(out), ptr, out.nbytes, stream))
        return out

    def synchronize(self, stream=None):
        _check(self.lib.silo_gpu_stream_synchronize(stream))

    # -- kernels --------------------------------------------------------------------------------
    def filter_eval(self, code, leaves, n_slots, out_bitset=None, out_count=None, stream=None):
        """code: flat list of uint32 (2 per instruction); leaves: list of device pointers (int)."""
        code = np.ascontiguousarray(code, dtype=np.uint32)
        leaf_array = (ctypes.c_void_p * max(1, len(leaves)))(*[
            (l.value if isinstance(l, ctypes.c_void_p) else l) for l in leaves
        ])
        prog = BitProg(len(code) // 2, code.ctypes.data_as(c_u32p), len(leaves), leaf_array, n_slots)
        _check(self.lib.silo_gpu_filter_eval(self.handle, ctypes.byref(prog), out_bitset, out_count, stream))

    def filter_eval_batch(self, programs, out_bitsets=None, stream=None):
        """programs: list of (code, leaves, n_slots); returns the cardinalities (one launch for all of them)."""
        return filter_eval_batch(self.handle, programs, out_bitsets, stream)

    def count_buffer(self, stream=None):
        """Zeroed accumulator for cardinalities: COUNT_SHARDS uint64 (their sum is the count)."""
        counter = self.malloc(8 * COUNT_SHARDS)
        self.memset(counter, 0, 8 * COUNT_SHARDS, stream)
        return counter

    def read_count(self, counter, stream=None):
        return int(self.read(counter, np.uint64, COUNT_SHARDS, stream).sum())

    def popcount(self, bitset, stream=None):
        counter = self.count_buffer(stream)
        _check(self.lib.silo_gpu_popcount(self.handle, bitset, counter, stream))
        value = self.read_count(counter, stream)
        self.free(counter)
        return value

    def mutations_scan_async(self, seqstore_id, filter_ptr, pos_begin, pos_end, counts_ptr, stream=None):
        _check(self.lib.silo_gpu_mutations_scan(self.handle, seqstore_id, filter_ptr, pos_begin, pos_end, counts_ptr, stream))

    def mutations_scan(self, seqstore_id, filter_ptr=None, pos_begin=0, pos_end=None, stream=None):
        """Returns uint32 counts [pos_end - pos_begin][n_scan_symbols]."""
        if pos_end is None:
            pos_end = self.positions(seqstore_id)
        n_scan = len(self.scan_symbols[seqstore_id])
        n = (pos_end - pos_begin) * n_scan
        counts = self.malloc(max(4, 4 * n))
        self.memset(counts, 0, max(4, 4 * n), stream)
        self.mutations_scan_async(seqstore_id, filter_ptr, pos_begin, pos_end, counts, stream)
        out = self.read(counts, np.uint32, n, stream).reshape(pos_end - pos_begin, n_scan)
        self.free(counts)
        return out

    def mutations_scan_batch(self, seqstore_id, filter_ptrs, pos_begin=0, pos_end=None, stream=None):
        """One pass over the planes for several filters; returns a list of uint32 count tables."""
        if pos_end is None:
            pos_end = self.positions(seqstore_id)
        n_scan = len(self.scan_symbols[seqstore_id])
        n = (pos_end - pos_begin) * n_scan
        outs = []
        for _ in filter_ptrs:
            buf = self.malloc(max(4, 4 * n))
            self.memset(buf, 0, max(4, 4 * n), stream)
            outs.append(buf)
        filters = (ctypes.c_void_p * len(filter_ptrs))(*[(f.value if isinstance(f, ctypes.c_void_p) else f) for f in filter_ptrs])
        counts = (ctypes.c_void_p * len(outs))(*[o.value for o in outs])
        _check(self.lib.silo_gpu_mutations_scan_batch(self.handle, seqstore_id, filters, len(filter_ptrs), pos_begin, pos_end, counts, stream))
        tables = [self.read(o, np.uint32, n, stream).reshape(pos_end - pos_begin, n_scan) for o in outs]
        for o in outs:
            self.free(o)
        return tables

    def scan_planes(self, seqstore_id):
        """Code planes per position of the store's most common layout: 2 (or 3) after the re-encoding at finalize, else 3 / 5."""
        return self.lib.silo_gpu_store_scan_planes(self.handle, seqstore_id)

    def scan_rows(self, seqstore_id, pos_begin, pos_end):
        """Plane rows the Mutations scan reads for positions [pos_begin, pos_end)."""
        return int(self.lib.silo_gpu_store_scan_rows(self.handle, seqstore_id, pos_begin, pos_end))

    def finalize_seqstore(self, seqstore_id):
        _check(self.lib.silo_gpu_store_finalize_seqstore(self.handle, seqstore_id))

    def scan_runs(self, seqstore_id):
        """Runs of the missing symbol a scan of the store reads (0 unless the store derives the most numerous symbol of its positions)."""
        return int(self.lib.silo_gpu_store_scan_runs(self.handle, seqstore_id))

    def scan_escapes(self, seqstore_id):
        return self.lib.silo_gpu_store_scan_escapes(self.handle, seqstore_id)

    def mutations_scan_ranges(self, ranges, filter_ptrs, stream=None):
        """Every filter over every (seqstore_id, pos_begin, pos_end) range in as few launches as the layouts allow;
        returns tables[range][filter]."""
        flat = (ctypes.c_uint32 * (3 * len(ranges)))(*[int(v) for r in ranges for v in r])  # silo_gpu_scan_range[]
        outs = []
        for seqstore_id, pos_begin, pos_end in ranges:
            n = (pos_end - pos_begin) * len(self.scan_symbols[seqstore_id])
            for _ in filter_ptrs:
                buf = self.malloc(max(4, 4 * n))
                self.memset(buf, 0, max(4, 4 * n), stream)
                outs.append(buf)
        filters = (ctypes.c_void_p * len(filter_ptrs))(*[(f.value if isinstance(f, ctypes.c_void_p) else f) for f in filter_ptrs])
        counts = (ctypes.c_void_p * max(1, len(outs)))(*[o.value for o in outs])
        _check(self.lib.silo_gpu_mutations_scan_ranges(self.handle, flat, len(ranges), filters, len(filter_ptrs), counts, stream))
        tables = []
        for r, (seqstore_id, pos_begin, pos_end) in enumerate(ranges):
            n_scan = len(self.scan_symbols[seqstore_id])
            tables.append([self.read(outs[r * len(filter_ptrs) + q], np.uint32, (pos_end - pos_begin) * n_scan, stream).reshape(pos_end - pos_begin, n_scan)
                           for q in range(len(filter_ptrs))])
        for o in outs:
            self.free(o)
        return tables

    # ---- metadata columns (K5 / K6) and FastaAligned ---------------------------------------------
    VALUE_TYPES = {np.dtype(np.int32): 0, np.dtype(np.uint32): 1, np.dtype(np.float64): 2}
    COMPARATORS = {"==": 0, "!=": 1, "<": 2, ">=": 3, ">": 4, "<=": 5}

    def upload_column(self, values):
        """int32 / uint32 / float64 array with one value per row -> device pointer (free with self.free)."""
        values = np.ascontiguousarray(values)
        out = ctypes.c_void_p()
        _check(self.lib.silo_gpu_upload_column(values.ctypes.data_as(ctypes.c_void_p), len(values), self.VALUE_TYPES[values.dtype], ctypes.byref(out)))
        return out

    def bitset_from_compare(self, column_ptr, dtype, comparator, value, stream=None):
        """Row bitset (as downloaded words) of `column <comparator> value`."""
        dtype = np.dtype(dtype)
        scalar = np.array([value], dtype=dtype)
        out = self.bitset_alloc()
        _check(self.lib.silo_gpu_bitset_from_compare(
            self.handle, out, column_ptr, self.VALUE_TYPES[dtype], self.COMPARATORS[comparator], scalar.ctypes.data_as(ctypes.c_void_p), stream))
        words = self.bitset_download(out, stream)
        self.free(out)
        return words

    def group_count(self, filter_ptr, id_ptrs, cardinalities, stream=None):
        """Histogram of the mixed-radix tuple ids (first column most significant) of the filtered rows."""
        n_bins = int(np.prod(cardinalities, dtype=np.int64))
        counts = self.malloc(4 * n_bins)
        self.memset(counts, 0, 4 * n_bins, stream)
        ids = (ctypes.c_void_p * len(id_ptrs))(*[p.value for p in id_ptrs])
        cards = (ctypes.c_uint32 * len(cardinalities))(*cardinalities)
        _check(self.lib.silo_gpu_group_count(self.handle, filter_ptr, ids, cards, len(id_ptrs), counts, stream))
        out = self.read(counts, np.uint32, n_bins, stream)
        self.free(counts)
        return out

    def group_count_hashed(self, filter_ptr, id_ptrs, cardinalities, max_rows, stream=None):
        """(tuple ids, counts) of the filtered rows through the HBM hash table (K6b), sorted by tuple id."""
        ids = (ctypes.c_void_p * len(id_ptrs))(*[p.value for p in id_ptrs])
        cards = (ctypes.c_uint32 * len(cardinalities))(*cardinalities)
        keys, counts, n = ctypes.c_void_p(), ctypes.c_void_p(), ctypes.c_uint32()
        _check(self.lib.silo_gpu_group_count_hashed(self.handle, filter_ptr, ids, cards, len(id_ptrs), max_rows, ctypes.byref(keys),
                                                    ctypes.byref(counts), ctypes.byref(n), stream))
        if n.value == 0:
            return np.zeros(0, np.uint64), np.zeros(0, np.uint32)
        tuple_ids = self.read(keys, np.uint64, n.value, stream)
        tuple_counts = self.read(counts, np.uint32, n.value, stream)
        self.free(keys)
        self.free(counts)
        order = np.argsort(tuple_ids)
        return tuple_ids[order], tuple_counts[order]

    def reconstruct_sequences(self, seqstore_id, rows, stream=None):
        """The stored characters of the given rows: uint8 array [len(rows)][positions]."""
        rows = np.ascontiguousarray(rows, dtype=np.uint32)
        positions = self.positions(seqstore_id)
        rows_dev = self.upload_column(rows)
        out = self.malloc(max(1, len(rows) * positions))
        _check(self.lib.silo_gpu_reconstruct_sequences(self.handle, seqstore_id, rows_dev, len(rows), out, stream))
        chars = self.read(out, np.uint8, len(rows) * positions, stream).reshape(len(rows), positions)
        self.free(out)
        self.free(rows_dev)
        return chars

    def mutations_select(self, counts, reference_index, min_proportion, capacity, stream=None):
        """K4 on a host-made count table [positions][symbols]: (n_selected, rows[min(n, capacity)] as (position, symbol, count, total))."""
        counts = np.ascontiguousarray(counts, dtype=np.uint32)
        positions, n_symbols = counts.shape
        counts_dev = self.upload_column(counts.reshape(-1))
        ref = np.ascontiguousarray(reference_index, dtype=np.uint8)
        ref_dev = ctypes.c_void_p()
        _check(self.lib.silo_gpu_upload_bytes(ref.ctypes.data_as(ctypes.c_void_p), ref.nbytes, ctypes.byref(ref_dev)))
        out = self.malloc(16 + 16 * max(capacity, 1))
        _check(self.lib.silo_gpu_mutations_select(counts_dev, ref_dev, positions, n_symbols, ctypes.c_double(min_proportion), capacity, out, stream))
        words = self.read(out, np.uint32, 4 + 4 * max(capacity, 1), stream)
        for pointer in (counts_dev, ref_dev, out):
            self.free(pointer)
        n = int(words[0])
        return n, words[4:4 + 4 * min(n, capacity)].reshape(-1, 4)

    def last_scan_kernel(self):
        return self.lib.silo_gpu_last_scan_kernel().decode()

    def tune(self, knob, value):
        return self.lib.silo_gpu_tune(knob, value)

    def build_pass(self, seqstore_id, which):
        """Two-pass build: 1 = the sequences that follow are only counted, 2 = they are written straight into the adaptive planes."""
        _check(self.lib.silo_gpu_store_build_pass(self.handle, seqstore_id, which))

    def build_mode(self, seqstore_id):
        return self.lib.silo_gpu_store_build_mode(self.handle, seqstore_id)
