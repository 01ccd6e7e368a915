"""ctypes wrapper of oracle/roaring_port.c — the CPU baseline ("port") and a third, independent checker.
TEST / BASELINE INFRASTRUCTURE ONLY (tests/ and bench.py's cpu_baseline leg)."""
import ctypes
import os

import numpy as np

from . import build as _build

_lib = None


class SynthT(ctypes.Structure):
    _fields_ = [
        ("seed", ctypes.c_uint64), ("n_sequences", ctypes.c_uint32), ("positions", ctypes.c_uint32), ("n_lineages", ctypes.c_uint32),
        ("lineage", ctypes.c_void_p), ("lead", ctypes.c_void_p), ("trail", ctypes.c_void_p), ("mstart", ctypes.c_void_p),
        ("mlen", ctypes.c_void_p), ("lineage_symbol", ctypes.c_void_p), ("reference", ctypes.c_void_p),
        ("private_threshold", ctypes.c_uint32), ("ambiguous_threshold", ctypes.c_uint32), ("is_aa", ctypes.c_uint32),
    ]


def load():
    global _lib
    if _lib is None:
        path = _build.build_all()[0]
        lib = ctypes.CDLL(path)
        vp = ctypes.c_void_p
        lib.port_store_build.argtypes = [vp, vp, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_uint32]
        lib.port_store_build.restype = vp
        lib.port_store_free.argtypes = [vp]
        lib.port_store_free.restype = None
        lib.port_filter_from_words.argtypes = [vp, ctypes.c_uint32]
        lib.port_filter_from_words.restype = vp
        lib.port_filter_free.argtypes = [vp]
        lib.port_filter_free.restype = None
        lib.port_filter_cardinality.argtypes = [vp]
        lib.port_filter_cardinality.restype = ctypes.c_uint64
        lib.port_and_cardinality.argtypes = [vp, vp]
        lib.port_and_cardinality.restype = ctypes.c_uint64
        lib.port_contains.argtypes = [vp, ctypes.c_uint32]
        lib.port_mutations_scan.argtypes = [vp, vp, vp, ctypes.c_int, ctypes.c_int]
        lib.port_mutations_scan.restype = ctypes.c_double
        lib.port_max_threads.restype = ctypes.c_int
        lib.port_store_census.argtypes = [vp, vp]
        lib.port_store_census.restype = None
        _lib = lib
    return _lib


def max_threads():
    return load().port_max_threads()


class Filter:
    def __init__(self, words, n_bits):
        self.lib = load()
        words = np.ascontiguousarray(words, dtype=np.uint64)
        assert len(words) * 64 >= n_bits
        self.handle = ctypes.c_void_p(self.lib.port_filter_from_words(words.ctypes.data_as(ctypes.c_void_p), n_bits))

    @property
    def cardinality(self):
        return int(self.lib.port_filter_cardinality(self.handle))

    def and_cardinality(self, other):
        return int(self.lib.port_and_cardinality(self.handle, other.handle))

    def contains(self, value):
        return bool(self.lib.port_contains(self.handle, value))

    def __del__(self):
        if getattr(self, "handle", None):
            self.lib.port_filter_free(self.handle)
            self.handle = None


class PortStore:
    """SequenceStorePartition over positions [pos_begin, pos_begin + n_positions) in roaring-format containers."""

    def __init__(self, n_sequences, pos_begin, n_positions, alphabet, model=None, symbols=None):
        self.lib = load()
        self.n_symbols = 16 if alphabet == "nuc" else 25
        self.n_positions = n_positions
        self._keep = []
        model_ptr = None
        symbols_ptr = None
        if model is not None:
            arrays = [
                np.ascontiguousarray(model.lineage_of_sequence, dtype=np.uint16), np.ascontiguousarray(model.lead_gap, dtype=np.uint32),
                np.ascontiguousarray(model.trail_gap, dtype=np.uint32), np.ascontiguousarray(model.missing_start, dtype=np.uint32),
                np.ascontiguousarray(model.missing_len, dtype=np.uint32), np.ascontiguousarray(model.lineage_symbol, dtype=np.uint8),
                np.ascontiguousarray(model.reference, dtype=np.uint8),
            ]
            self._keep += arrays
            desc = SynthT(
                model.seed, n_sequences, model.positions, model.n_lineages, *[a.ctypes.data for a in arrays],
                model.private_threshold, model.ambiguous_threshold, 0 if alphabet == "nuc" else 1)
            self._keep.append(desc)
            model_ptr = ctypes.cast(ctypes.pointer(desc), ctypes.c_void_p)
        else:
            symbols = np.ascontiguousarray(symbols, dtype=np.uint8)
            assert symbols.shape == (n_sequences, n_positions)
            self._keep.append(symbols)
            symbols_ptr = symbols.ctypes.data_as(ctypes.c_void_p)
        self.handle = ctypes.c_void_p(self.lib.port_store_build(
            model_ptr, symbols_ptr, n_sequences, pos_begin, n_positions, 0 if alphabet == "nuc" else 1))

    def mutations_scan(self, filter_=None, n_threads=0, grain=300):
        """Returns (counts uint32 [n_positions][n_symbols], seconds)."""
        counts = np.zeros((self.n_positions, self.n_symbols), dtype=np.uint32)
        seconds = self.lib.port_mutations_scan(
            self.handle, None if filter_ is None else filter_.handle, counts.ctypes.data_as(ctypes.c_void_p), n_threads, grain)
        return counts, float(seconds)

    def census(self):
        out = np.zeros(4, dtype=np.uint64)
        self.lib.port_store_census(self.handle, out.ctypes.data_as(ctypes.c_void_p))
        return dict(arrays=int(out[0]), bitsets=int(out[1]), runs=int(out[2]), bytes=int(out[3]))

    def close(self):
        if getattr(self, "handle", None):
            self.lib.port_store_free(self.handle)
            self.handle = None

    def __del__(self):
        self.close()
