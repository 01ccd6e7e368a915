"""Writes tests/golden/roaring/vectors.json: portable-format roaring serializations as oracle/roaring_format.py produces them from
the published specification (array, bitset and run containers; with and without run containers and the offset header), each with
the count and a SHA-256 of its sorted ids (as little-endian uint32).  The reference holds no serialized state among its fixtures and
CRoaring is not in the image, so these vectors pin the readers (oracle/roaring_format.deserialize and
silo_gpu_store_import_position) against the WRITER HERE, not against CRoaring: parity unpinned.
usage: python tests/golden/make_roaring_fixtures.py"""
import hashlib
import json
import os
import random
import struct
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import roaring_format  # noqa: E402

rng = random.Random(20250117)
cases = {
    "empty": [],
    "single": [7],
    "chunk_boundary": [65535, 65536, 131071, 131072],
    "array": sorted(rng.sample(range(65536), 900)),
    "bitset": sorted(rng.sample(range(65536), 9000)),
    "one_run": list(range(100, 60000)),
    "runs_and_arrays_3_containers": list(range(10, 5000)) + [70000, 70001] + sorted(rng.sample(range(131072, 196608), 300)),
    "runs_and_bitset_5_containers": (list(range(0, 40000)) + sorted(rng.sample(range(65536, 131072), 7000)) + list(range(140000, 140010))
                                     + [200000] + list(range(262144, 300000))),
}
vectors = []
for name, ids in cases.items():
    for use_runs in (True, False):
        payload = roaring_format.serialize(ids, use_runs=use_runs)
        assert roaring_format.deserialize(payload) == sorted(set(ids))
        digest = hashlib.sha256(struct.pack(f"<{len(ids)}I", *sorted(ids))).hexdigest()
        vectors.append({"name": name, "use_runs": use_runs, "n": len(ids), "ids_sha256": digest, "payload_hex": payload.hex()})
out = os.path.join(ROOT, "tests", "golden", "roaring")
os.makedirs(out, exist_ok=True)
json.dump({"_source": "oracle/roaring_format.py (published RoaringFormatSpec restated); parity unpinned", "vectors": vectors},
          open(os.path.join(out, "vectors.json"), "w"), indent=0)
print("wrote", len(vectors), "vectors")
