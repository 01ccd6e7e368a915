"""Naive per-character checker over a symbol matrix — test infrastructure only.

The second, representation-free oracle of SURVEY.md §7 step 1: it never builds a bitmap index, it
counts characters.  Used to pin both the reference-shaped oracle (oracle/silo_oracle.py) and the HIP
kernels on the same inputs.
"""
import numpy as np


def pack_bits(mask):
    """bool [N] -> uint64 words, bit i of word w = row 64*w + i (little endian)."""
    mask = np.asarray(mask, dtype=bool)
    n_words = (len(mask) + 63) // 64
    padded = np.zeros(n_words * 64, dtype=bool)
    padded[: len(mask)] = mask
    return np.packbits(padded, bitorder="little").view("<u8").copy()


def unpack_bits(words, n):
    words = np.ascontiguousarray(words, dtype="<u8")
    return np.unpackbits(words.view(np.uint8), bitorder="little")[:n].astype(bool)


def plane(symbols, position, symbol):
    """Membership set C[p][s] of SURVEY.md §3.6 as packed words."""
    return pack_bits(symbols[:, position] == symbol)


def mutation_counts(symbols, filter_mask, scan_symbols, pos_begin=0, pos_end=None):
    """uint32 [positions][len(scan_symbols)]: |F ∧ C[p][s]| by direct counting (mutations.cpp:64-164)."""
    symbols = np.asarray(symbols)
    if pos_end is None:
        pos_end = symbols.shape[1]
    selected = symbols[np.asarray(filter_mask, dtype=bool), pos_begin:pos_end]
    out = np.zeros((pos_end - pos_begin, len(scan_symbols)), dtype=np.uint32)
    for k, s in enumerate(scan_symbols):
        out[:, k] = (selected == s).sum(axis=0, dtype=np.int64)
    return out
