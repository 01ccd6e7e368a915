"""CPU-side checks: the native libraries load and export every declared symbol, the bit-program
instruction semantics (shared header, g++ build), and the product's loud failure without a GPU."""
import ctypes
import os
import re

import numpy as np
import pytest

from oracle import dense

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_functions(header):
    text = open(os.path.join(ROOT, "include", header)).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(silo_(?:gpu|engine)_[a-z0-9_]+)\s*\(", text)))


def test_libsilo_gpu_exports_every_declared_symbol(built):
    from silo_amd import binding

    lib = binding.load_library()
    declared = [name for name in declared_functions("silo_gpu.h") if not name.endswith("_desc")]
    assert len(declared) >= 25
    for name in declared:
        assert hasattr(lib, name), name
    assert sorted(binding.EXPORTED_SYMBOLS) == sorted(declared)


def test_libsilo_engine_exports_every_declared_symbol(built):
    from silo_amd import engine

    lib = engine.load_library()
    declared = [n for n in declared_functions("silo_engine.h") if n.startswith("silo_engine_") and n != "silo_engine_all_reduce_u32"]
    for name in declared:
        assert hasattr(lib, name), name
    assert sorted(engine.EXPORTED_SYMBOLS) == sorted(declared)


def test_product_fails_loudly_without_a_gpu(built):
    import subprocess
    import sys

    # run in a child so that a visible GPU on the test box does not change the outcome
    code = (
        "import sys; sys.path[:0]=[%r, %r]\n"
        "import numpy as np\n"
        "from silo_amd import binding\n"
        "try:\n"
        "    binding.GpuStore(8, [dict(name='m', alphabet='nuc', reference=np.ones(4, dtype=np.uint8))])\n"
        "    print('CREATED')\n"
        "except binding.SiloGpuError as e:\n"
        "    print('ERR', e.code)\n"
    ) % (ROOT, os.path.join(ROOT, "lapis-silo_amd"))
    env = dict(os.environ, HIP_VISIBLE_DEVICES="-1", ROCR_VISIBLE_DEVICES="")
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, timeout=120).stdout
    assert "ERR -4" in out, out  # SILO_GPU_ERR_NO_DEVICE: there is no CPU fallback


@pytest.fixture(scope="module")
def bitprog(built):
    lib = ctypes.CDLL(os.path.join(ROOT, "lapis-silo_amd", "lib", "libbitprog_host.so"))
    lib.bitprog_eval_host.argtypes = [ctypes.c_void_p, ctypes.c_uint32, ctypes.c_void_p, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_void_p]

    def run(code, masks, n):
        n_words = (n + 63) // 64 + 1  # one padding word: must come out zero
        code = np.ascontiguousarray(code, dtype=np.uint32)
        leaves = np.zeros((max(1, len(masks)), n_words), dtype=np.uint64)
        for k, mask in enumerate(masks):
            words = dense.pack_bits(mask)
            leaves[k, : len(words)] = words
        out = np.zeros(n_words, dtype=np.uint64)
        lib.bitprog_eval_host(code.ctypes.data, len(code) // 2, leaves.ctypes.data, len(masks), n_words, n, out.ctypes.data)
        assert out[-1] == 0
        return dense.unpack_bits(out, n)

    return run


@pytest.mark.parametrize("n", [1, 64, 65, 1000])
def test_bitprog_semantics(bitprog, n):
    from silo_amd import binding as b

    rng = np.random.default_rng(n)
    masks = [rng.random(n) < p for p in (0.5, 0.3, 0.8, 0.1, 0.6, 0.5, 0.9)]
    full = np.ones(n, bool)
    assert np.array_equal(bitprog(b.encode(b.OP_ONES, 0), [], n), full)
    assert np.array_equal(bitprog(b.encode(b.OP_ZERO, 0) + b.encode(b.OP_NOT, 0, 0), [], n), full)
    code = (b.encode(b.OP_LOAD, 1, imm=0) + b.encode(b.OP_LOAD, 2, imm=1) + b.encode(b.OP_ANDNOT, 3, 1, 2)
            + b.encode(b.OP_LOAD, 1, imm=2) + b.encode(b.OP_OR, 3, 3, 1) + b.encode(b.OP_MOV, 0, 3))
    assert np.array_equal(bitprog(code, masks, n), (masks[0] & ~masks[1]) | masks[2])
    # n-ary forms over runs of consecutive leaves (imm = first | count << 16), incl. runs longer than one batch of 8
    many = masks + [rng.random(n) < 0.97 for _ in range(13)]
    for first, count in [(0, 1), (0, 7), (2, 5), (0, 8), (0, 9), (3, 17), (0, 20)]:
        run = many[first:first + count]
        imm = first | (count << 16)
        assert np.array_equal(bitprog(b.encode(b.OP_OR_N, 0, imm=imm), many, n), np.logical_or.reduce(run)), (first, count)
        assert np.array_equal(bitprog(b.encode(b.OP_AND_N, 0, imm=imm), many, n), np.logical_and.reduce(run)), (first, count)
        zero = b.encode(b.OP_ZERO, 1) + b.encode(b.OP_ZERO, 2) + b.encode(b.OP_ZERO, 3) + b.encode(b.OP_ZERO, 4) + b.encode(b.OP_ZERO, 5)
        total_run = sum(m.astype(int) for m in run)
        for k in (0, 1, count // 2, count, count + 1):
            code = zero + b.encode(b.OP_CNT_ADD_N, 1, 0, 5, imm=imm) + b.encode(b.OP_CNT_GE, 0, 1, 5, imm=k)
            assert np.array_equal(bitprog(code, many, n), total_run >= k), (first, count, k)
            code = zero + b.encode(b.OP_CNT_ADD_NOT_N, 1, 0, 5, imm=imm) + b.encode(b.OP_CNT_EQ, 0, 1, 5, imm=k)
            assert np.array_equal(bitprog(code, many, n), (count - total_run) == k), (first, count, k)
    total = sum(m.astype(int) for m in masks)
    for k in range(0, 10):
        code = b.encode(b.OP_ZERO, 1) + b.encode(b.OP_ZERO, 2) + b.encode(b.OP_ZERO, 3)
        for leaf in range(7):
            code += b.encode(b.OP_LOAD, 0, imm=leaf) + b.encode(b.OP_CNT_ADD, 1, 0, 3)
        assert np.array_equal(bitprog(code + b.encode(b.OP_CNT_GE, 0, 1, 3, imm=k), masks, n), total >= k), k
        assert np.array_equal(bitprog(code + b.encode(b.OP_CNT_EQ, 0, 1, 3, imm=k), masks, n), total == k), k


# ---- pure host functions of the engine against the oracle's restatement (no device) ---------------------------------
@pytest.fixture(scope="module")
def host_logic(built):
    import json

    path = os.path.join(ROOT, "lapis-silo_amd", "lib", "libhost_logic.so")
    if not os.path.exists(path):
        pytest.fail("libhost_logic.so has not been built")
    lib = ctypes.CDLL(path)
    lib.t_string_to_date.restype = ctypes.c_uint32
    lib.t_string_to_date.argtypes = [ctypes.c_char_p]
    lib.t_date_to_string.argtypes = [ctypes.c_uint32, ctypes.c_char_p, ctypes.c_size_t]
    lib.t_lineage.argtypes = [ctypes.c_char_p, ctypes.c_char_p, ctypes.c_int, ctypes.c_char_p, ctypes.c_size_t]
    lib.t_insertion_standardise.argtypes = [ctypes.c_char_p, ctypes.c_char_p, ctypes.c_char_p, ctypes.c_size_t]
    lib.t_describe_fasta.argtypes = [ctypes.c_char_p, ctypes.c_char_p, ctypes.c_size_t]
    lib.t_describe_database_config.argtypes = [ctypes.c_char_p, ctypes.c_int, ctypes.c_char_p, ctypes.c_size_t]
    lib.alias_json = open(os.path.join(ROOT, "tests", "golden", "exampleDataset", "pangolineage_alias.json")).read().encode()
    lib.alias_dict = json.loads(lib.alias_json)
    return lib


def test_dates_match_oracle(host_logic):
    from oracle import silo_oracle as so

    buffer = ctypes.create_string_buffer(64)
    texts = ["2021-03-18", "2020-1-1", "", "2021", "2021-13-01", "2021-00-10", "2021-02-31", "2021-02-32", "x-y-z", "2021-03-18T10:00",
             "0001-01-01", "99999-12-31", "2021-3", " 2021-03-04", "2021-03-4x", "-5-03-04"]
    for text in texts:
        date = host_logic.t_string_to_date(text.encode())
        assert date == so.string_to_date(text), text
        n = host_logic.t_date_to_string(date, buffer, 64)
        assert (buffer.value.decode() if n >= 0 else None) == so.date_to_string(date), text


def test_lineage_aliases_match_oracle(host_logic):
    from oracle import silo_oracle as so
    from tests import dataset

    lookup = so.PangoLineageAliasLookup(host_logic.alias_dict)
    lineages = sorted({row["pango_lineage"] for row in dataset.load_example_dataset()["rows"]})
    lineages += ["B.1.617.2.43", "B.1.617.2", "AY.43", "XA.1", "X", "Q.1", "BA.5.2.1.7", "B.1.1.529.5.2.1.7", "nonsense", "", "AY", "B.1.1.7.", "Q.3 . 4"]
    buffer = ctypes.create_string_buffer(256)
    for lineage in lineages:
        unaliased = lookup.unalias(lineage)
        assert host_logic.t_lineage(host_logic.alias_json, lineage.encode(), 0, buffer, 256) >= 0
        assert buffer.value.decode() == unaliased, lineage
        assert host_logic.t_lineage(host_logic.alias_json, unaliased.encode(), 1, buffer, 256) >= 0
        assert buffer.value.decode() == lookup.alias(unaliased), lineage
        assert host_logic.t_lineage(host_logic.alias_json, lineage.encode(), 2, buffer, 256) >= 0
        assert buffer.value.decode() == lookup.alias(unaliased), lineage


def test_insertion_values_are_standardised_like_the_reference(host_logic):
    """insertion_column.cpp:31-113: entries without a sequence name belong to the default sequence and are written
    back without it; quotes are dropped; malformed entries are errors."""
    buffer = ctypes.create_string_buffer(512)
    cases = [
        (b"main", "22204:CAGAA", "22204:CAGAA"), (b"main", "main:22204:CAGAA", "22204:CAGAA"), (b"main", "other:5:AC,7:G", "other:5:AC,7:G"),
        (b"main", '"25701:CCC"', "25701:CCC"), (None, "S:214:EPE,ORF1a:3:T", "S:214:EPE,ORF1a:3:T"), (b"main", "", ""),
        (None, "214:EPE", None), (b"main", "a:b", None), (b"main", "1:2:3:4", None), (b"main", "S:x:EPE", None), (b"main", "99999999999:A", None),
    ]
    for default_sequence, value, expected in cases:
        n = host_logic.t_insertion_standardise(default_sequence, value.encode(), buffer, 512)
        assert (buffer.value.decode() if n >= 0 else None) == expected, (default_sequence, value)


def test_reference_unit_test_vectors_on_the_host_functions(host_logic):
    """date.test.cpp, pango_lineage_alias.test.cpp and insertion_column.test.cpp vectors against the C++ host code."""
    import json

    vectors = json.load(open(os.path.join(ROOT, "tests", "golden", "operators", "operator_vectors.json")))
    buffer = ctypes.create_string_buffer(256)
    for text, value in vectors["dates"]["parse"]:
        assert host_logic.t_string_to_date(text.encode()) == value, text
    for text, printed in vectors["dates"]["reprint"]:
        n = host_logic.t_date_to_string(host_logic.t_string_to_date(text.encode()), buffer, 256)
        assert (buffer.value.decode() if n >= 0 else None) == printed, text
    alias = vectors["lineage_alias"]
    for mode, block in ((0, alias["unalias"]), (1, alias["alias"])):
        alias_json = json.dumps(block["alias_key"]).encode()
        for text, expected in block["cases"]:
            assert host_logic.t_lineage(alias_json, text.encode(), mode, buffer, 256) >= 0
            assert buffer.value.decode() == expected, (mode, text)
    for text, expected in alias["example_file"]:
        assert host_logic.t_lineage(host_logic.alias_json, text.encode(), 0, buffer, 256) >= 0
        assert buffer.value.decode() == expected, text


def test_database_config_reader_and_validation_vectors(host_logic):
    """database_config.test.cpp (the reference's own YAML fixtures through the loader's reader) and
    config_repository.test.cpp (its validation rules, with the messages that test expects)."""
    import json

    directory = os.path.join(ROOT, "tests", "golden", "config")
    buffer = ctypes.create_string_buffer(8192)
    for case in json.load(open(os.path.join(directory, "config_vectors.json")))["cases"]:
        status = host_logic.t_describe_database_config(os.path.join(directory, case["file"]).encode(), int(case["validate"]), buffer, 8192)
        text = buffer.value.decode()
        if case["error"] is not None:
            assert status == -1 and case["error"] in text, (case["cite"], text)
            continue
        assert status >= 0, (case["cite"], text)
        got = json.loads(text)
        for key, value in case.get("expect", {}).items():
            if key == "metadata":
                assert [[c["name"], c["type"], c["generateIndex"]] for c in got["metadata"]] == value, case["cite"]
            else:
                assert got[key] == value, (case["cite"], key)


def test_fasta_reader_vectors(host_logic):
    """fasta_reader.test.cpp on the reference's own fixture files, through the loader's FASTA reader."""
    import json

    directory = os.path.join(ROOT, "tests", "golden", "fasta")
    buffer = ctypes.create_string_buffer(4096)
    for case in json.load(open(os.path.join(directory, "fasta_vectors.json")))["cases"]:
        status = host_logic.t_describe_fasta(os.path.join(directory, case["file"]).encode(), buffer, 4096)
        text = buffer.value.decode()
        if "error" in case:
            assert status == -1 and case["error"] in text, (case["cite"], text)
        else:
            assert status >= 0 and json.loads(text) == case["records"], (case["cite"], text)


# ---- the layout of the adaptive planes (csrc/layout_choice.h: host code of the device library) -----------------------------
def choose_layouts(host_logic, totals, n_bits, sequences, one_hot=1, key_cost=0):
    """one_hot: 0 = code planes only, 1 = one-hot rows (a row for every stored symbol), 2 = the most numerous symbol derived."""
    import ctypes

    import numpy as np

    totals = np.ascontiguousarray(totals, dtype=np.uint32)
    positions, n_scan = totals.shape
    code_map = np.zeros((positions, 8), dtype=np.uint8)
    escapes = np.zeros((positions, n_scan), dtype=np.uint32)
    host_logic.t_choose_layouts.argtypes = [ctypes.c_void_p, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_uint64, ctypes.c_int, ctypes.c_uint64,
                                            ctypes.c_void_p, ctypes.c_void_p]
    host_logic.t_choose_layouts.restype = None
    host_logic.t_choose_layouts(totals.ctypes.data, n_scan, n_bits, positions, sequences, int(one_hot), key_cost, code_map.ctypes.data, escapes.ctypes.data)
    rows = code_map[:, 0] & 0x0F
    kind = np.where(code_map[:, 0] & 0x80, "identity", np.where(code_map[:, 0] & 0x20, "derived", np.where(code_map[:, 0] & 0x40, "one-hot", "codes")))
    return rows, kind, code_map, escapes


def test_layout_choice_follows_the_cost_model(host_logic):
    """One row where one symbol has nearly every row, two where a second symbol is frequent, 2 code planes where three are
    over a stretch long enough to pay for a launch of its own, identity planes where nothing else pays; every row a position
    does not store is an escape key."""
    import numpy as np

    n = 10_000_000
    settled = [n - 2000, 900, 600, 300, 200]           # one symbol, 2 000 other rows
    two = [n // 2, n // 2 - 3000, 1500, 1000, 500]     # two frequent symbols
    three = [n // 3, n // 3, n // 3 - 600, 400, 200]   # three
    flat = [n // 5] * 5                                # all five
    totals = np.array([settled] * 40 + [two] + [settled] * 40 + [three] * 300 + [flat] * 12 + [settled] * 20, dtype=np.uint32)
    rows, kind, code_map, escapes = choose_layouts(host_logic, totals, 3, n)
    assert list(kind[:81]) == ["one-hot"] * 81 and list(rows[:40]) == [1] * 40 and rows[40] == 2 and list(rows[41:81]) == [1] * 40
    assert list(kind[81:381]) == ["codes"] * 300 and list(rows[81:381]) == [2] * 300     # a run of its own: cheaper than 3 rows each, launch included
    assert list(kind[381:393]) == ["identity"] * 12 and list(rows[381:393]) == [3] * 12
    assert list(kind[393:]) == ["one-hot"] * 20
    assert code_map[0, 1] == 0 and code_map[40, 1] == 0 and code_map[40, 2] == 1 and list(code_map[90, 1:4]) == [0, 1, 2]
    assert list(escapes[0]) == [0, 900, 600, 300, 200] and list(escapes[40]) == [0, 0, 1500, 1000, 500]
    assert list(escapes[90]) == [0, 0, 0, 400, 200] and not escapes[381:393].any()
    # a few positions with three frequent symbols between settled ones: not worth a launch of their own (192 MB of plane bytes)
    totals = np.array([settled] * 10 + [three] * 30 + [settled] * 10, dtype=np.uint32)
    rows, kind, _, _ = choose_layouts(host_logic, totals, 3, n)
    assert list(kind) == ["one-hot"] * 50 and list(rows[10:40]) == [3] * 30
    # without one-hot rows (SILO_GPU_TUNE_COMPACT_INDEX 2): two code planes everywhere
    rows, kind, _, escapes = choose_layouts(host_logic, totals, 3, n, one_hot=0)
    assert list(kind) == ["codes"] * 50 and list(rows) == [2] * 50 and list(escapes[0]) == [0, 0, 0, 300, 200]
    # the key cost moves the boundary between one row and two: 40 000 rows of a second symbol are keys at 16 B, a row at 40 B
    second = [n - 41000, 40000, 500, 300, 200]
    assert choose_layouts(host_logic, np.array([second] * 8, dtype=np.uint32), 3, n)[0].tolist() == [1] * 8
    assert choose_layouts(host_logic, np.array([second] * 8, dtype=np.uint32), 3, n, key_cost=40)[0].tolist() == [2] * 8


def test_layout_choice_derives_the_most_numerous_symbol(host_logic):
    """Mode 2 (a store that keeps its missing symbol as runs): the most numerous symbol of a one-hot position gets no row and
    no keys — code_map[p][7] names it — and the rows are those of the symbols behind it: none at a settled position, one
    where a second symbol is frequent, two where three are; the escape keys are the symbols with neither."""
    import numpy as np

    n = 10_000_000
    settled = [n - 2000, 900, 600, 300, 200]
    two = [3000, n // 2, n // 2 - 3000 - 2500, 1500, 1000]   # the second and third symbol are the frequent ones
    three = [n // 3, n // 3, n // 3 - 600, 400, 200]
    empty = [0, 0, 0, 0, 0]                                  # no row has a valid symbol here (all missing)
    totals = np.array([settled] * 5 + [two] + [three] * 2 + [empty] + [settled] * 5, dtype=np.uint32)
    rows, kind, code_map, escapes = choose_layouts(host_logic, totals, 3, n, one_hot=2)
    assert list(kind) == ["derived"] * 14
    assert list(rows) == [0] * 5 + [1] + [2, 2] + [0] + [0] * 5
    assert code_map[0, 7] == 0 and list(escapes[0]) == [0, 900, 600, 300, 200]
    assert code_map[5, 7] == 1 and code_map[5, 1] == 2 and list(escapes[5]) == [3000, 0, 0, 1500, 1000]
    assert code_map[6, 7] == 0 and list(code_map[6, 1:3]) == [1, 2] and list(escapes[6]) == [0, 0, 0, 400, 200]
    assert code_map[8, 7] == 0 and not escapes[8].any()     # derived count there: |filter| - rows without a valid symbol = 0
    # amino acids (5 identity planes): a saturated stretch long enough keeps its identity planes, the rest derives
    aa_settled = [n - 5000] + [250] * 20 + [0]
    aa_flat = [n // 22] * 22
    totals = np.array([aa_settled] * 20 + [aa_flat] * 40 + [aa_settled] * 20, dtype=np.uint32)
    rows, kind, _, _ = choose_layouts(host_logic, totals, 5, n, one_hot=2)
    assert list(kind[:20]) == ["derived"] * 20 and list(rows[:20]) == [0] * 20 and list(kind[20:60]) == ["identity"] * 40 and list(rows[20:60]) == [5] * 40


def test_layout_choice_on_the_real_alignment(host_logic):
    """The 1 000 real SARS-CoV-2 sequences of exampleDataset1000Sequences, counted per position and scaled to 10 M rows:
    the layout choice of the device library gives them 1.03 plane rows per position — what the synthetic model gets."""
    import numpy as np

    from tests import test_config1_dataset

    _, _, sequences = test_config1_dataset.load()
    rows_ = [s for s in sequences if s is not None]
    matrix = np.frombuffer("".join(rows_).encode(), dtype=np.uint8).reshape(len(rows_), -1)
    totals = np.stack([(matrix == ord(c)).sum(axis=0) for c in "-ACGT"], axis=1).astype(np.uint32) * 10_000
    derived_rows, derived_kind, _, derived_escapes = choose_layouts(host_logic, totals, 3, 10_000_000, one_hot=2)
    print(f"with the most numerous symbol derived: plane rows per position {derived_rows.sum() / len(derived_rows):.4f}, escape keys {int(derived_escapes.sum()) / 1e6:.1f} M")
    assert derived_rows.sum() < 0.06 * len(derived_rows) and np.mean(derived_kind == "derived") > 0.999
    rows, kind, _, escapes = choose_layouts(host_logic, totals, 3, 10_000_000)
    per_position = rows.sum() / len(rows)
    keys = int(escapes.sum())
    print(f"plane rows per position {per_position:.4f}; one-hot {np.mean(kind == 'one-hot'):.4f}, codes {np.mean(kind == 'codes'):.4f}, "
          f"identity {np.mean(kind == 'identity'):.4f}; escape keys {keys / 1e6:.1f} M = {keys / totals.sum():.2e} of the cells")
    assert 1.0 <= per_position < 1.06 and np.mean(rows == 1) > 0.95 and np.mean(kind == "identity") < 0.001
    assert keys < 0.002 * totals.sum()


# ---- the row order of the loader (host/dataset_loader.cpp) ------------------------------------------------------------------
def test_loader_orders_rows_as_the_reference_does(host_logic):
    """preprocessor.cpp:159-227 + database_config.cpp:190-198: rows by partitionBy key, then dateToSortBy, then primary key; rows
    without a date last; without a partitionBy column by date and key alone; stable."""
    import ctypes

    def order(partition_keys, dates, primary_keys):
        n = len(primary_keys)
        as_array = lambda values: None if values is None else (ctypes.c_char_p * n)(*[v.encode() for v in values])
        out = (ctypes.c_uint32 * n)()
        host_logic.t_reference_row_order.restype = None
        host_logic.t_reference_row_order(as_array(partition_keys), as_array(dates), as_array(primary_keys), n, out)
        return list(out)

    lineages = ["B.1.1.7", "B.1", "B.1.1.7", "AY.4", "B.1", "B.1.1.7", "AY.4"]
    dates = ["2021-03-01", "2020-11-05", "2021-01-10", "", "2020-11-05", "2021-01-10", "2021-08-20"]
    keys = ["k6", "k5", "k4", "k3", "k2", "k1", "k0"]
    assert order(lineages, dates, keys) == [6, 3, 4, 1, 5, 2, 0]   # AY.4 (dated, then undated), B.1 (same date: by key), B.1.1.7 by date then key
    assert order(None, dates, keys) == [4, 1, 5, 2, 0, 6, 3]       # one partition: date, key; the undated row last
    assert order(None, None, keys) == [6, 5, 4, 3, 2, 1, 0]
    assert order(lineages, None, ["same"] * 7) == [3, 6, 1, 4, 0, 2, 5]  # ties keep the input order
