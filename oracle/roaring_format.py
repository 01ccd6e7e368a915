"""CRoaring's PORTABLE serialization of a bitmap, written and read from its published specification
(github.com/RoaringBitmap/RoaringFormatSpec) — the payload the reference's saveDatabaseState writes per symbol and position
(include/silo/roaring/roaring_serialize.h:17-45: size_t length + Roaring::write bytes).  CRoaring 1.0.0 itself is not in the
image (conanfile.py:16, not vendored), so this restatement is pinned by nothing but the specification: PARITY UNPINNED.
TEST INFRASTRUCTURE ONLY: it makes the fixtures for silo_gpu_store_import_position / _missing_rows (tests/) and reads them back.

Layout: uint32 cookie — low 16 bits 12347: run containers may occur, n = (cookie >> 16) + 1, then ceil(n / 8) bytes of run flags;
12346: no run containers, then uint32 n — then n x (uint16 key, uint16 cardinality - 1), then n x uint32 data offsets (absent when
cookie 12347 and n < 4), then per container: array (cardinality <= 4096): uint16 values ascending; bitset: 1024 x uint64;
run: uint16 n_runs, n_runs x (uint16 start, uint16 length - 1).  All little-endian.
"""
import struct

SERIAL_COOKIE_NO_RUNCONTAINER = 12346
SERIAL_COOKIE = 12347
NO_OFFSET_THRESHOLD = 4


def _runs(values):
    runs = []
    start = previous = values[0]
    for value in values[1:]:
        if value != previous + 1:
            runs.append((start, previous - start))
            start = value
        previous = value
    runs.append((start, previous - start))
    return runs


def serialize(ids, use_runs=True):
    """ids: iterable of non-negative ints < 2**32 -> bytes.  Container choice as CRoaring's runOptimize makes it: a run container where
    it is the smallest of the three encodings (use_runs=False: arrays and bitsets only, the 12346 cookie)."""
    chunks = {}
    for value in sorted(set(int(i) for i in ids)):
        chunks.setdefault(value >> 16, []).append(value & 0xFFFF)
    containers = []
    for key in sorted(chunks):
        values = chunks[key]
        cardinality = len(values)
        plain_size = 2 * cardinality if cardinality <= 4096 else 8192
        runs = _runs(values)
        if use_runs and 2 + 4 * len(runs) < plain_size:
            data = struct.pack("<H", len(runs)) + b"".join(struct.pack("<HH", start, length) for start, length in runs)
            containers.append((key, cardinality, True, data))
        elif cardinality <= 4096:
            containers.append((key, cardinality, False, struct.pack(f"<{cardinality}H", *values)))
        else:
            words = [0] * 1024
            for value in values:
                words[value >> 6] |= 1 << (value & 63)
            containers.append((key, cardinality, False, struct.pack("<1024Q", *words)))
    n = len(containers)
    any_run = any(is_run for _, _, is_run, _ in containers)
    if any_run:
        flags = bytearray((n + 7) // 8)
        for k, (_, _, is_run, _) in enumerate(containers):
            if is_run:
                flags[k // 8] |= 1 << (k % 8)
        header = struct.pack("<I", SERIAL_COOKIE | ((n - 1) << 16)) + bytes(flags)
    else:
        header = struct.pack("<II", SERIAL_COOKIE_NO_RUNCONTAINER, n)
    header += b"".join(struct.pack("<HH", key, cardinality - 1) for key, cardinality, _, _ in containers)
    has_offsets = (not any_run) or n >= NO_OFFSET_THRESHOLD
    offset = len(header) + (4 * n if has_offsets else 0)
    offsets = b""
    for _, _, _, data in containers:
        if has_offsets:
            offsets += struct.pack("<I", offset)
        offset += len(data)
    return header + offsets + b"".join(data for _, _, _, data in containers)


def deserialize(payload):
    """bytes -> sorted list of ints (the reader twin of serialize; raises ValueError on a malformed payload)."""
    if len(payload) < 8:
        raise ValueError("payload shorter than its header")
    (cookie,) = struct.unpack_from("<I", payload, 0)
    cursor = 4
    flags = None
    if cookie & 0xFFFF == SERIAL_COOKIE:
        n = (cookie >> 16) + 1
        flags = payload[cursor:cursor + (n + 7) // 8]
        cursor += (n + 7) // 8
    elif cookie == SERIAL_COOKIE_NO_RUNCONTAINER:
        (n,) = struct.unpack_from("<I", payload, cursor)
        cursor += 4
    else:
        raise ValueError("unknown cookie")
    descriptors = [struct.unpack_from("<HH", payload, cursor + 4 * k) for k in range(n)]
    cursor += 4 * n
    has_offsets = flags is None or n >= NO_OFFSET_THRESHOLD
    offsets = None
    if has_offsets:
        offsets = [struct.unpack_from("<I", payload, cursor + 4 * k)[0] for k in range(n)]
        cursor += 4 * n
    ids = []
    for k, (key, cardinality_minus_one) in enumerate(descriptors):
        cardinality = cardinality_minus_one + 1
        data = offsets[k] if has_offsets else cursor
        base = key << 16
        if flags is not None and (flags[k // 8] >> (k % 8)) & 1:
            (n_runs,) = struct.unpack_from("<H", payload, data)
            for r in range(n_runs):
                start, length = struct.unpack_from("<HH", payload, data + 2 + 4 * r)
                ids.extend(range(base + start, base + start + length + 1))
            cursor = data + 2 + 4 * n_runs
        elif cardinality <= 4096:
            ids.extend(base + v for v in struct.unpack_from(f"<{cardinality}H", payload, data))
            cursor = data + 2 * cardinality
        else:
            words = struct.unpack_from("<1024Q", payload, data)
            for w, word in enumerate(words):
                while word:
                    low = word & -word
                    ids.append(base + 64 * w + low.bit_length() - 1)
                    word ^= low
            cursor = data + 8192
    return ids
