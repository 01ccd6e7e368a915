"""Parity of the HIP kernels (through the C ABI) against the naive per-character oracle."""
import ctypes

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from oracle import dense  # noqa: E402
from oracle import synth as oracle_synth  # noqa: E402

NUC_CHARS = np.frombuffer(b"-ACGTRYSWKMBDHVN", dtype=np.uint8)
AA_CHARS = np.frombuffer(b"-ACDEFGHIKLMNPQRSTVWYBZ*X", dtype=np.uint8)


def random_symbols(rng, n, positions, alphabet):
    """Random alignment with a skewed symbol distribution (mostly the 'reference')."""
    if alphabet == "nuc":
        probs = np.array([0.05, 0.3, 0.2, 0.2, 0.15] + [0.005] * 10 + [0.05])
    else:
        probs = np.array([0.04] + [0.04] * 20 + [0.01, 0.01, 0.04, 0.1])
    probs = probs / probs.sum()
    return rng.choice(len(probs), size=(n, positions), p=probs).astype(np.uint8)


def make_store(n, stores):
    from silo_amd.binding import GpuStore

    return GpuStore(n, stores)


@pytest.mark.parametrize("n", [1, 63, 64, 100, 1000, 4097])
def test_transpose_planes_match_naive(built, n):
    rng = np.random.default_rng(n)
    positions = 37
    sym = random_symbols(rng, n, positions, "nuc")
    is_null = (rng.random(n) < 0.05).astype(np.uint8)
    sym_effective = sym.copy()
    sym_effective[is_null.astype(bool)] = 15
    ref = rng.integers(1, 5, size=positions).astype(np.uint8)
    with make_store(n, [dict(name="main", alphabet="nuc", reference=ref)]) as store:
        # two batches with an unaligned boundary exercise the atomicOr path
        cut = n // 3
        store.append_sequences(0, 0, NUC_CHARS[sym[:cut]], is_null[:cut])
        store.append_sequences(0, cut, NUC_CHARS[sym[cut:]], is_null[cut:])
        store.finalize()
        for p in [0, 1, positions // 2, positions - 1]:
            for s in range(16):
                got = store.plane_download(0, p, s)
                want = dense.pack_bits(sym_effective[:, p] == s)
                assert np.array_equal(got[: len(want)], want), (p, s)
                assert not got[len(want):].any()


def test_transpose_rejects_illegal_character(built):
    from silo_amd.binding import SiloGpuError

    ref = np.ones(8, dtype=np.uint8)
    with make_store(4, [dict(name="main", alphabet="nuc", reference=ref)]) as store:
        with pytest.raises(SiloGpuError):
            store.append_sequences(0, 0, ["ACGTACGT", "ACGTACGJ", "ACGTACGT", "ACGTACGT"])


@pytest.mark.parametrize("n,alphabet", [(100, "nuc"), (100, "aa"), (5000, "nuc"), (70000, "aa"), (140000, "nuc"), (300001, "nuc")])
def test_mutations_scan_matches_naive(built, n, alphabet):
    rng = np.random.default_rng(n + 7)
    positions = 23
    sym = random_symbols(rng, n, positions, alphabet)
    chars = (NUC_CHARS if alphabet == "nuc" else AA_CHARS)[sym]
    ref = sym[0].copy()
    with make_store(n, [dict(name="s", alphabet=alphabet, reference=ref)]) as store:
        store.append_sequences(0, 0, chars)
        store.finalize()
        scan_symbols = list(store.scan_symbols[0])
        filt = rng.random(n) < 0.37
        fptr = store.bitset_alloc()
        store.bitset_upload(fptr, dense.pack_bits(filt))
        for variant in (0, 1, 2):
            store.tune(1, variant)
            got = store.mutations_scan(0, fptr)
            want = dense.mutation_counts(sym, filt, scan_symbols)
            assert np.array_equal(got, want), (variant, store.last_scan_kernel())
            # sub-range + full filter
            got = store.mutations_scan(0, None, 3, 11)
            want = dense.mutation_counts(sym, np.ones(n, bool), scan_symbols, 3, 11)
            assert np.array_equal(got, want)
        store.tune(1, 0)
        assert store.popcount(fptr) == int(filt.sum())


def _eval(store, code, leaves, n_slots):
    from silo_amd.binding import encode  # noqa: F401

    out = store.bitset_alloc()
    cnt = store.count_buffer()
    store.filter_eval(code, leaves, n_slots, out, cnt)
    words = store.bitset_download(out)
    count = store.read_count(cnt)
    store.free(out)
    store.free(cnt)
    return words, count


@pytest.mark.parametrize("n", [100, 777, 100000])
def test_filter_eval_ops(built, n):
    from silo_amd import binding as b

    rng = np.random.default_rng(n)
    masks = [rng.random(n) < p for p in (0.5, 0.2, 0.7, 0.05, 0.9)]
    ref = np.ones(4, dtype=np.uint8)
    with make_store(n, [dict(name="s", alphabet="nuc", reference=ref)]) as store:
        leaves = []
        for m in masks:
            ptr = store.bitset_alloc()
            store.bitset_upload(ptr, dense.pack_bits(m))
            leaves.append(ptr)
        full = np.ones(n, bool)

        def check(code, n_slots, want):
            words, count = _eval(store, code, leaves, n_slots)
            assert np.array_equal(dense.unpack_bits(words, n), want)
            assert not dense.unpack_bits(words, store.row_words * 64)[n:].any()
            assert count == int(want.sum())

        # (m0 & m1) | ~m2, minus m3
        code = (b.encode(b.OP_LOAD, 0, imm=0) + b.encode(b.OP_LOAD, 1, imm=1) + b.encode(b.OP_AND, 0, 0, 1)
                + b.encode(b.OP_LOAD, 1, imm=2) + b.encode(b.OP_NOT, 1, 1) + b.encode(b.OP_OR, 0, 0, 1)
                + b.encode(b.OP_LOAD, 1, imm=3) + b.encode(b.OP_ANDNOT, 0, 0, 1))
        check(code, 2, ((masks[0] & masks[1]) | ~masks[2]) & ~masks[3])
        # leaf operands: (leaf0 & leaf1) | ~leaf2, minus leaf3 without a single LOAD
        L = b.LEAF_OPERAND
        code = (b.encode(b.OP_AND, 0, L + 0, L + 1) + b.encode(b.OP_NOT, 1, L + 2) + b.encode(b.OP_OR, 0, 0, 1)
                + b.encode(b.OP_ANDNOT, 0, 0, L + 3))
        check(code, 2, ((masks[0] & masks[1]) | ~masks[2]) & ~masks[3])
        check(b.encode(b.OP_MOV, 0, L + 4), 1, masks[4])
        # n-ary runs: Or / And over leaves 0..4 and 1..3, 2-of-5 and "exactly 1 of the complements"
        check(b.encode(b.OP_OR_N, 0, imm=0 | (5 << 16)), 1, np.logical_or.reduce(masks))
        check(b.encode(b.OP_AND_N, 0, imm=1 | (3 << 16)), 1, masks[1] & masks[2] & masks[3])
        zero3 = b.encode(b.OP_ZERO, 1) + b.encode(b.OP_ZERO, 2) + b.encode(b.OP_ZERO, 3)
        total5 = sum(m.astype(int) for m in masks)
        check(zero3 + b.encode(b.OP_CNT_ADD_N, 1, 0, 3, imm=0 | (5 << 16)) + b.encode(b.OP_CNT_GE, 0, 1, 3, imm=2), 4, total5 >= 2)
        check(zero3 + b.encode(b.OP_CNT_ADD_NOT_N, 1, 0, 3, imm=0 | (5 << 16)) + b.encode(b.OP_CNT_EQ, 0, 1, 3, imm=1), 4, (5 - total5) == 1)
        check(b.encode(b.OP_ONES, 0), 1, full)
        check(b.encode(b.OP_ZERO, 0), 1, ~full)
        check(b.encode(b.OP_ZERO, 0) + b.encode(b.OP_NOT, 0, 0), 1, full)
        # n-of-5 thresholds with a 3-bit counter in slots 1..3
        total = sum(m.astype(int) for m in masks)
        for k in range(0, 7):
            code = b.encode(b.OP_ZERO, 1) + b.encode(b.OP_ZERO, 2) + b.encode(b.OP_ZERO, 3)
            for leaf in range(5):
                if leaf % 2 == 0:
                    code += b.encode(b.OP_LOAD, 0, imm=leaf) + b.encode(b.OP_CNT_ADD, 1, 0, 3)
                else:
                    code += b.encode(b.OP_CNT_ADD, 1, b.LEAF_OPERAND + leaf, 3)
            check(code + b.encode(b.OP_CNT_GE, 0, 1, 3, imm=k), 4, total >= k)
            check(code + b.encode(b.OP_CNT_EQ, 0, 1, 3, imm=k), 4, total == k)


def test_synthetic_store_matches_cpu_twin(built):
    from silo_amd import synth

    n, positions = 3000, 211
    tree = synth.make_lineage_tree(40)
    lineage = synth.assign_lineages(n, tree, 1234)
    for k, alphabet in enumerate(["nuc", "aa"]):
        ref = synth.random_reference(positions, alphabet, 99 + k)
        model = synth.make_model(n, ref, alphabet, tree, lineage, seed=4321, store_index=k)
        # raise the ambiguity rate so the sparse path is exercised at this size
        model.ambiguous_threshold = 1 << 14
        sym = oracle_synth.symbol_matrix(model, np.arange(n), np.arange(positions))
        with make_store(n, [dict(name="s", alphabet=alphabet, reference=ref)]) as store:
            store.generate_synthetic(0, model)
            store.finalize()
            n_symbols = 16 if alphabet == "nuc" else 25
            for p in [0, 1, 57, 100, positions - 1]:
                for s in range(n_symbols):
                    got = store.plane_download(0, p, s)
                    want = dense.pack_bits(sym[:, p] == s)
                    assert np.array_equal(got[: len(want)], want), (alphabet, p, s)
            filt = tree.subtree(1)[lineage].astype(bool)
            fptr = store.bitset_alloc()
            store.bitset_from_lineages(fptr, tree.subtree(1))
            assert np.array_equal(dense.unpack_bits(store.bitset_download(fptr), n), filt)
            got = store.mutations_scan(0, fptr)
            want = dense.mutation_counts(sym, filt, list(store.scan_symbols[0]))
            assert np.array_equal(got, want)


@pytest.mark.parametrize("n,alphabet", [(5000, "nuc"), (140000, "aa"), (300001, "nuc")])
def test_batched_scan_matches_single_scans(built, n, alphabet):
    """K1c: one pass over the planes for 1..9 filters gives exactly the tables of the single-filter scans."""
    rng = np.random.default_rng(n + 3)
    positions = 19
    sym = random_symbols(rng, n, positions, alphabet)
    chars = (NUC_CHARS if alphabet == "nuc" else AA_CHARS)[sym]
    with make_store(n, [dict(name="s", alphabet=alphabet, reference=sym[0].copy())]) as store:
        store.append_sequences(0, 0, chars)
        store.finalize()
        scan_symbols = list(store.scan_symbols[0])
        masks = [rng.random(n) < p for p in (0.5, 0.01, 0.9, 0.3, 0.0, 1.0, 0.6, 0.2, 0.75)]
        ptrs = []
        for mask in masks:
            ptr = store.bitset_alloc()
            store.bitset_upload(ptr, dense.pack_bits(mask))
            ptrs.append(ptr)
        for count in (1, 2, 3, 4, 5, 6, 7, 8, 9):
            tables = store.mutations_scan_batch(0, ptrs[:count], 2, 17)
            for mask, table in zip(masks, tables):
                assert np.array_equal(table, dense.mutation_counts(sym, mask, scan_symbols, 2, 17)), count


@pytest.mark.parametrize("n,alphabet", [(70000, "aa"), (140000, "nuc"), (300001, "nuc")])
def test_sparse_filters_take_the_gather_scan_and_give_the_same_tables(built, n, alphabet):
    """K1s: filters with set bits in few 64-byte sectors are routed on the device to k_scan_gather; dense ones, and the filters just
    past the capacity, stay with k_scan_sliced — same tables either way, alone and mixed in one batch."""
    rng = np.random.default_rng(n + 11)
    positions = 23
    sym = random_symbols(rng, n, positions, alphabet)
    chars = (NUC_CHARS if alphabet == "nuc" else AA_CHARS)[sym]
    with make_store(n, [dict(name="s", alphabet=alphabet, reference=sym[0].copy())]) as store:
        store.append_sequences(0, 0, chars)
        store.finalize()
        scan_symbols = list(store.scan_symbols[0])
        capacity = max(4, store.row_words // 16)  # 64-byte sectors (8 words) of the filter with a set bit

        def rows_in_sectors(n_sectors):
            mask = np.zeros(n, bool)
            sectors = rng.choice(n // 512, size=n_sectors, replace=False)
            mask[sectors * 512 + rng.integers(0, 512, size=n_sectors)] = True
            return mask

        masks = [np.zeros(n, bool), rows_in_sectors(1), rows_in_sectors(7), rows_in_sectors(capacity), rows_in_sectors(capacity + 1),
                 rng.random(n) < 0.4]
        last = np.zeros(n, bool)
        last[n - 1] = True  # the last row, in the ragged last word
        masks.append(last)
        clustered = np.zeros(n, bool)
        clustered[n // 3: n // 3 + 500] = True
        masks.append(clustered)
        ptrs = []
        for mask in masks:
            ptr = store.bitset_alloc()
            store.bitset_upload(ptr, dense.pack_bits(mask))
            ptrs.append(ptr)
        want = [dense.mutation_counts(sym, mask, scan_symbols) for mask in masks]
        for divisor in (0, -1):  # default routing, then the dense kernels alone
            store.tune(3, divisor)
            for ptr, table in zip(ptrs, want):
                assert np.array_equal(store.mutations_scan(0, ptr), table), divisor
            for got, table in zip(store.mutations_scan_batch(0, ptrs, 0, positions), want):
                assert np.array_equal(got, table), divisor
            got = store.mutations_scan(0, ptrs[2], 2, 17)
            assert np.array_equal(got, dense.mutation_counts(sym, masks[2], scan_symbols, 2, 17))
        store.tune(3, 0)


def skewed_symbols(rng, n, positions, alphabet):
    """Alignment-like columns: per position a dominant symbol (97.2 %), a second (1.5 %) and a third (0.8 %) valid symbol,
    0.03 % spread over the remaining valid symbols, 0.5 % missing / ambiguous."""
    from silo_amd import alphabet as alphabets

    table = alphabets.NUCLEOTIDE if alphabet == "nuc" else alphabets.AMINO_ACID
    valid = np.array(list(table.valid_mutation_symbols))
    others = np.array([s for s in range(table.count) if s not in set(valid.tolist())])
    out = np.empty((n, positions), dtype=np.uint8)
    for p in range(positions):
        order = rng.permutation(valid)
        probs = np.zeros(table.count)
        probs[order[0]], probs[order[1]], probs[order[2]] = 0.9717, 0.015, 0.008
        probs[order[3:]] = 0.0003 / (len(valid) - 3)
        probs[others] = 0.005 / len(others)
        out[:, p] = rng.choice(table.count, size=n, p=probs / probs.sum())
    return out


def settle_positions(rng, sym, columns, alphabet, second=0.0002):
    """Columns where ONE valid symbol has nearly every row, a second valid symbol `second` of them, and the remaining valid
    symbols 0.01 %, the missing symbol and the ambiguity codes 0.1 %: what most positions of a real alignment look like."""
    from silo_amd import alphabet as alphabets

    table = alphabets.NUCLEOTIDE if alphabet == "nuc" else alphabets.AMINO_ACID
    valid = np.array(list(table.valid_mutation_symbols))
    others = np.array([s for s in range(table.count) if s not in set(valid.tolist())])
    for p in columns:
        order = rng.permutation(valid)
        probs = np.zeros(table.count)
        probs[order[0]], probs[order[1]] = 0.9989 - second, second
        probs[order[2:]] = 0.0001 / (len(valid) - 2)
        probs[others] = 0.001 / len(others)
        sym[:, p] = rng.choice(table.count, size=len(sym), p=probs / probs.sum())


@pytest.mark.parametrize("n,alphabet", [(140000, "nuc"), (70000, "aa"), (300001, "nuc"), (1200003, "nuc")])  # the last: 10 slices of escape keys
def test_adaptive_code_planes_give_the_same_tables(built, n, alphabet):
    """finalize() re-encodes alignment-like data per position into one-hot rows (0-3 of them: the most numerous symbol is
    derived, knob 0; 1-3 with a row for it too, knob 3) or 2 code planes (3 symbols) + escape keys, and releases the build-time
    planes; knob 2 leaves the one-hot rows out, with the re-encoding switched off the store keeps its 3 / 5 identity planes.
    All four stores answer every scan (single, batched, sub-ranges, sparse
    filters, several ranges), every filter leaf and FastaAligned like the naive oracle."""
    rng = np.random.default_rng(n + 17)
    positions = 29
    sym = skewed_symbols(rng, n, positions, alphabet)
    sym2 = skewed_symbols(rng, n, 11, alphabet)
    # one position where many symbols are frequent: it keeps more planes than its neighbours (a run of its own)
    table_size = 16 if alphabet == "nuc" else 25
    sym[:, 13] = rng.integers(0, table_size, size=n)
    # settled positions: one row each as one-hot rows (an odd number of rows in the run), one of them with a second frequent symbol
    settled = list(range(3, 10)) + list(range(19, 27))
    settle_positions(rng, sym, settled, alphabet)
    settle_positions(rng, sym, [22], alphabet, second=0.02)
    settle_positions(rng, sym2, range(11), alphabet)  # a store that is ONE run of one-hot rows
    chars = NUC_CHARS if alphabet == "nuc" else AA_CHARS
    sparse = np.zeros(n, bool)
    sparse[rng.choice(n, size=9, replace=False)] = True
    masks = [rng.random(n) < 0.4, sparse, np.ones(n, bool), rng.random(n) < 0.02, np.zeros(n, bool)]
    # a clustered filter (rows in lineage or date order): whole column tiles and key slices without a selected row are skipped
    masks[3] = (np.arange(n) >= int(n * 0.6)) & (rng.random(n) < 0.7)
    sizes = {}
    # re-encoded with the most numerous symbol of a position derived (0), with a one-hot row for it too (3), without one-hot rows
    # (2), then the identity planes kept (-1); the charge per kind of launch is off (knob 9: these rows are short, with it a
    # store would never mix layouts)
    for knob in (0, 3, 2, -1):
        with make_store(n, [dict(name="a", alphabet=alphabet, reference=sym[0].copy()), dict(name="b", alphabet=alphabet, reference=sym2[0].copy())]) as store:
            store.tune(4, knob)
            store.tune(9, -1)
            try:
                store.append_sequences(0, 0, chars[sym])
                store.append_sequences(1, 0, chars[sym2])
                store.finalize()
            finally:
                store.tune(4, 0)
                store.tune(9, 0)
            sizes[knob] = store.device_bytes
            scan_symbols = list(store.scan_symbols[0])
            full_planes = 3 if alphabet == "nuc" else 5
            if knob == 2:
                assert store.scan_planes(0) == 2 and store.scan_planes(1) == 2
                rows = store.scan_rows(0, 0, positions)
                assert 2 * positions < rows <= 2 * (positions - 1) + full_planes  # position 13 keeps more planes
                assert store.scan_rows(0, 13, 14) > 2 and store.scan_rows(0, 12, 13) == 2
                assert 0 < store.scan_escapes(0) <= n * positions // 200
                with pytest.raises(Exception):  # the build-time planes are gone
                    store.append_sequences(0, 0, chars[sym[:1]])
            elif knob == 0:
                assert store.scan_rows(1, 0, 11) == 0 and store.scan_planes(1) == 0 and store.scan_runs(1) > 0  # positions without a row
                assert store.scan_rows(0, 3, 10) == 0 and store.scan_rows(0, 19, 27) == 1 and store.scan_rows(0, 22, 23) == 1
                assert store.scan_rows(0, 13, 14) > 1 and store.scan_rows(0, 12, 13) >= 1 and 3 <= store.scan_rows(0, 0, 3) <= 6
                assert 0 < store.scan_escapes(0) <= n * positions // 200
                assert store.scan_escapes(1) > 0
            elif knob == 3:
                assert store.scan_rows(1, 0, 11) == 11 and store.scan_planes(1) == 1 and store.scan_runs(1) == 0
                assert store.scan_rows(0, 3, 10) == 7 and store.scan_rows(0, 19, 27) == 9 and store.scan_rows(0, 22, 23) == 2
                assert store.scan_rows(0, 13, 14) > 2 and store.scan_rows(0, 12, 13) == 2 and store.scan_rows(0, 0, 3) == 6
                assert 0 < store.scan_escapes(0) <= n * positions // 200
                assert store.scan_escapes(1) > 0  # every non-dominant valid symbol of store 1 is a key
            else:
                assert store.scan_planes(0) == full_planes and store.scan_rows(0, 0, positions) == full_planes * positions
                assert store.scan_escapes(0) == 0
            ptrs = []
            for mask in masks:
                ptr = store.bitset_alloc()
                store.bitset_upload(ptr, dense.pack_bits(mask))
                ptrs.append(ptr)
            want = [dense.mutation_counts(sym, mask, scan_symbols) for mask in masks]
            want2 = [dense.mutation_counts(sym2, mask, scan_symbols) for mask in masks]
            assert int(want[2].sum()) == int(np.isin(sym, scan_symbols).sum())
            for ptr, table in zip(ptrs, want):
                assert np.array_equal(store.mutations_scan(0, ptr), table), knob
            assert np.array_equal(store.mutations_scan(0, None), want[2])
            for got, table in zip(store.mutations_scan_batch(0, ptrs, 0, positions), want):
                assert np.array_equal(got, table), knob
            # the escape pass on the caller's stream (2) and over the position-major keys (3: what stores of more than 512 slices use)
            for mode in (2, 3):
                store.tune(5, mode)
                try:
                    assert np.array_equal(store.mutations_scan(0, ptrs[0]), want[0]) and np.array_equal(store.mutations_scan(0, ptrs[3], 2, 27), want[3][2:27]), (knob, mode)
                    for got, table in zip(store.mutations_scan_batch(0, ptrs[:3], 0, positions), want):
                        assert np.array_equal(got, table), (knob, mode)
                finally:
                    store.tune(5, 0)
            assert np.array_equal(store.mutations_scan(0, ptrs[0], 3, 20), dense.mutation_counts(sym, masks[0], scan_symbols, 3, 20))
            tables = store.mutations_scan_ranges([(0, 0, positions), (1, 0, 11), (0, 5, 6), (0, 13, 14), (0, 12, 15), (0, 8, 23), (1, 4, 6)], ptrs[:3])
            for q in range(3):
                assert np.array_equal(tables[0][q], want[q]) and np.array_equal(tables[1][q], want2[q])
                assert np.array_equal(tables[2][q], want[q][5:6]) and np.array_equal(tables[3][q], want[q][13:14])
                assert np.array_equal(tables[4][q], want[q][12:15]) and np.array_equal(tables[5][q], want[q][8:23])
                assert np.array_equal(tables[6][q], want2[q][4:6])
            for got, table in zip(store.mutations_scan_batch(1, ptrs, 0, 11), want2):
                assert np.array_equal(got, table), knob
            # every symbol's one-hot plane (filter leaves): coded symbols, escape keys, the missing symbol, sparse symbols
            for position in (0, 5, 13, 22, positions - 1):
                for symbol in range(table_size):
                    got = store.plane_download(0, position, symbol)
                    want_plane = dense.pack_bits(sym[:, position] == symbol)
                    assert np.array_equal(got[: len(want_plane)], want_plane), (knob, position, symbol)
                    assert not got[len(want_plane):].any()
            # FastaAligned reads the same planes (and the keys)
            picked = np.concatenate([rng.choice(n, size=40, replace=False), np.nonzero(~np.isin(sym[:, 13], scan_symbols))[0][:5]]).astype(np.uint32)
            assert np.array_equal(store.reconstruct_sequences(0, picked), chars[sym[picked]])
    assert sizes[0] < sizes[3] < sizes[2] < 0.85 * sizes[-1]  # <= 2 planes + the missing-symbol plane instead of 3 / 5 + 1


def test_scan_timings_cover_every_plane_row(built):
    """silo_gpu_scan_timings (what bench.py's roofline of the dominant launch rests on): with SILO_GPU_TUNE_SCAN_TIMING set,
    a scan reports one entry per k_scan_sliced launch, named as rocprofv3 names the kernel, and the plane rows of the
    entries add up to the rows the store holds; without the knob nothing is recorded."""
    from silo_amd import binding

    n, positions = 140000, 29
    rng = np.random.default_rng(5)
    sym = skewed_symbols(rng, n, positions, "nuc")
    settle_positions(rng, sym, range(3, 17), "nuc")
    with make_store(n, [dict(name="a", alphabet="nuc", reference=sym[0].copy())]) as store:
        store.append_sequences(0, 0, NUC_CHARS[sym])
        store.tune(4, 3)   # a one-hot row for the most numerous symbol too, and
        store.tune(9, -1)  # no charge per kind of launch: this small store is to mix one-hot rows and code planes
        try:
            store.finalize()
        finally:
            store.tune(4, 0)
            store.tune(9, 0)
        mask = rng.random(n) < 0.4
        ptr = store.bitset_alloc()
        store.bitset_upload(ptr, dense.pack_bits(mask))
        want = dense.mutation_counts(sym, mask, list(store.scan_symbols[0]))
        store.tune(7, 1)
        try:
            assert np.array_equal(store.mutations_scan(0, ptr), want)
            entries = binding.scan_timings()
        finally:
            store.tune(7, 0)
        planes = [e for e in entries if e["kernel"].startswith("k_scan_sliced<")]
        assert len(planes) >= 2  # one-hot rows and 2 code planes at least
        assert all(e["ms"] > 0 and e["blocks"] > 0 and e["filters"] == 1 and e["bytes"] > 0 for e in entries)
        assert any(e["kernel"] == "k_scan_sliced<2, 2, 8, 1, 2>" for e in planes)
        assert sum(e["plane_rows"] for e in planes) == store.scan_rows(0, 0, positions)
        row_bytes = 8 * (((n + 63) // 64 + 31) // 32 * 32)
        assert all(e["bytes"] == (e["plane_rows"] + 1) * row_bytes for e in planes)  # the launch's plane rows + the filter row
        # the escape-key pass is a launch of the scan as well: 8 bytes per key (+ the filter slices of its blocks)
        escapes = [e for e in entries if e["kernel"].startswith("k_scan_escapes_sliced<")]
        assert len(escapes) == 1 and escapes[0]["bytes"] >= 8 * store.scan_escapes(0) > 0
        assert np.array_equal(store.mutations_scan(0, ptr), want)
        assert binding.scan_timings() == []


@pytest.mark.parametrize("n", [900, 140000])
def test_scan_over_several_ranges_in_one_call(built, n):
    """silo_gpu_mutations_scan_ranges: every filter over ranges of nucleotide and amino-acid stores of different lengths
    (more than 16 ranges, so more than one launch per layout), dense and sparse filters mixed."""
    rng = np.random.default_rng(n + 13)
    lengths = [("nuc", 23), ("aa", 7), ("nuc", 40), ("aa", 31)] + [("aa", 3 + i) for i in range(17)]
    syms = [random_symbols(rng, n, positions, alphabet) for alphabet, positions in lengths]
    stores = [dict(name=f"s{i}", alphabet=alphabet, reference=sym[0].copy()) for i, ((alphabet, _), sym) in enumerate(zip(lengths, syms))]
    with make_store(n, stores) as store:
        for i, ((alphabet, _), sym) in enumerate(zip(lengths, syms)):
            store.append_sequences(i, 0, (NUC_CHARS if alphabet == "nuc" else AA_CHARS)[sym])
        store.finalize()
        sparse = np.zeros(n, bool)
        sparse[rng.choice(n, size=5, replace=False)] = True
        masks = [rng.random(n) < 0.3, sparse, rng.random(n) < 0.8, np.zeros(n, bool), sparse | (rng.random(n) < 0.001)]
        ptrs = []
        for mask in masks:
            ptr = store.bitset_alloc()
            store.bitset_upload(ptr, dense.pack_bits(mask))
            ptrs.append(ptr)
        ranges = [(i, 0, positions) for i, (_, positions) in enumerate(lengths)]
        ranges[0] = (0, 2, 17)
        ranges[3] = (3, 30, 31)
        ranges.append((2, 5, 5))  # empty
        for n_filters in (1, 3, 5):
            tables = store.mutations_scan_ranges(ranges, ptrs[:n_filters])
            for (seqstore_id, pos_begin, pos_end), per_filter in zip(ranges, tables):
                for mask, table in zip(masks, per_filter):
                    want = dense.mutation_counts(syms[seqstore_id], mask, list(store.scan_symbols[seqstore_id]), pos_begin, pos_end)
                    assert np.array_equal(table, want), (seqstore_id, n_filters)


@pytest.mark.parametrize("n", [1, 64, 1000, 250007])
def test_column_compare_matches_numpy(built, n):
    """K5: every comparator on int32 / uint32 / float64 columns, NULL markers and NaN included."""
    rng = np.random.default_rng(n)
    ref = np.ones(4, dtype=np.uint8)
    ints = rng.integers(-5, 60, size=n).astype(np.int32)
    ints[rng.random(n) < 0.05] = np.iinfo(np.int32).min
    words = rng.integers(0, 1 << 30, size=n).astype(np.uint32)
    words[rng.random(n) < 0.3] = 7
    floats = rng.choice([0.9, 0.93, 0.98, -0.0, 0.0, 1e300], size=n)
    floats[rng.random(n) < 0.1] = np.nan
    operators = {"==": np.equal, "!=": np.not_equal, "<": np.less, ">=": np.greater_equal, ">": np.greater, "<=": np.less_equal}
    with make_store(n, [dict(name="s", alphabet="nuc", reference=ref)]) as store:
        for column, probes in ((ints, [0, 30, np.iinfo(np.int32).min, -3]), (words, [7, 0, 1 << 29]), (floats, [0.93, 0.0, np.nan, 2.0])):
            pointer = store.upload_column(column)
            for probe in probes:
                for name, function in operators.items():
                    with np.errstate(invalid="ignore"):
                        want = function(column, column.dtype.type(probe))
                    got = store.bitset_from_compare(pointer, column.dtype, name, probe)
                    assert np.array_equal(dense.unpack_bits(got, n), want), (column.dtype, name, probe)
                    assert not dense.unpack_bits(got, store.row_words * 64)[n:].any()
            store.free(pointer)


@pytest.mark.parametrize("n", [1, 777, 300001])
def test_group_count_matches_numpy(built, n):
    """K6: LDS-privatised and global histograms, skewed and uniform ids, with and without a filter."""
    rng = np.random.default_rng(n + 1)
    ref = np.ones(4, dtype=np.uint8)
    with make_store(n, [dict(name="s", alphabet="nuc", reference=ref)]) as store:
        filt = rng.random(n) < 0.4
        fptr = store.bitset_alloc()
        store.bitset_upload(fptr, dense.pack_bits(filt))
        for cardinalities in ([3], [60, 50], [2, 3, 5, 7], [5000], [300, 200]):
            columns = []
            for cardinality in cardinalities:
                ids = rng.integers(0, cardinality, size=n).astype(np.uint32)
                ids[rng.random(n) < 0.7] = cardinality // 2  # one dominant group
                columns.append(ids)
            pointers = [store.upload_column(ids) for ids in columns]
            key = np.zeros(n, dtype=np.int64)
            for ids, cardinality in zip(columns, cardinalities):
                key = key * cardinality + ids
            n_bins = int(np.prod(cardinalities))
            assert np.array_equal(store.group_count(fptr, pointers, cardinalities), np.bincount(key[filt], minlength=n_bins))
            assert np.array_equal(store.group_count(None, pointers, cardinalities), np.bincount(key, minlength=n_bins))
            for pointer in pointers:
                store.free(pointer)


@pytest.mark.parametrize("n", [5, 4097, 200003])
def test_group_count_hashed_matches_numpy(built, n):
    """K6b: tuple spaces far beyond the dense histogram (here up to 2^50 potential tuples)."""
    rng = np.random.default_rng(n + 2)
    ref = np.ones(4, dtype=np.uint8)
    with make_store(n, [dict(name="s", alphabet="nuc", reference=ref)]) as store:
        filt = rng.random(n) < 0.6
        fptr = store.bitset_alloc()
        store.bitset_upload(fptr, dense.pack_bits(filt))
        for cardinalities in ([1 << 20, 1 << 20, 1 << 10], [n + 7, 3], [40000, 50000]):
            columns = []
            for cardinality in cardinalities:
                ids = rng.integers(0, min(cardinality, 5000), size=n).astype(np.uint32) * (cardinality // min(cardinality, 5000))
                ids[rng.random(n) < 0.3] = cardinality - 1
                columns.append(ids.astype(np.uint32))
            pointers = [store.upload_column(ids) for ids in columns]
            key = np.zeros(n, dtype=np.uint64)
            for ids, cardinality in zip(columns, cardinalities):
                key = key * np.uint64(cardinality) + ids.astype(np.uint64)
            for mask, pointer in ((filt, fptr), (np.ones(n, bool), None)):
                want_ids, want_counts = np.unique(key[mask], return_counts=True)
                got_ids, got_counts = store.group_count_hashed(pointer, pointers, cardinalities, int(mask.sum()))
                assert np.array_equal(got_ids, want_ids) and np.array_equal(got_counts, want_counts.astype(np.uint32))
            for pointer in pointers:
                store.free(pointer)


def test_reconstruct_sequences_matches_input(built):
    """FastaAligned gather: every stored character comes back, IUPAC codes (sparse planes) and null genomes included."""
    rng = np.random.default_rng(5)
    n, positions = 3000, 97
    for alphabet, chars_of in (("nuc", NUC_CHARS), ("aa", AA_CHARS)):
        sym = random_symbols(rng, n, positions, alphabet)
        is_null = (rng.random(n) < 0.02).astype(np.uint8)
        with make_store(n, [dict(name="s", alphabet=alphabet, reference=sym[0].copy())]) as store:
            store.append_sequences(0, 0, chars_of[sym], is_null)
            store.finalize()
            expected = chars_of[sym].copy()
            expected[is_null.astype(bool)] = ord("N" if alphabet == "nuc" else "X")
            rows = np.concatenate([[0, n - 1], rng.choice(n, size=200, replace=False), np.nonzero(is_null)[0][:5]]).astype(np.uint32)
            got = store.reconstruct_sequences(0, rows)
            assert np.array_equal(got, expected[rows])


def test_selection_unit_test_vectors_on_the_compare_kernel(built):
    """selection.test.cpp:10-113 through K5, the negation as the negated comparator (selection.cpp:195-220)."""
    import json
    import os

    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "operators", "operator_vectors.json")
    negated = {"==": "!=", "!=": "==", "<": ">=", ">=": "<", ">": "<=", "<=": ">"}
    for vec in json.load(open(path))["selection"]:
        column = np.array(vec["column"], dtype=np.int32)
        with make_store(len(column), [dict(name="s", alphabet="nuc", reference=np.ones(4, dtype=np.uint8))]) as store:
            pointer = store.upload_column(column)
            got = dense.unpack_bits(store.bitset_from_compare(pointer, np.int32, vec["comparator"], vec["value"]), len(column))
            assert np.nonzero(got)[0].tolist() == vec["expected"], vec["cite"]
            got = dense.unpack_bits(store.bitset_from_compare(pointer, np.int32, negated[vec["comparator"]], vec["value"]), len(column))
            assert np.nonzero(got)[0].tolist() == vec["negated"], vec["cite"]
            store.free(pointer)


def test_group_count_hashed_reports_an_understated_row_bound(built):
    """max_rows smaller than the rows the filter selects: an error status, not a hang or an out-of-bounds write."""
    from silo_amd.binding import SiloGpuError

    n = 50000
    rng = np.random.default_rng(9)
    with make_store(n, [dict(name="s", alphabet="nuc", reference=np.ones(4, dtype=np.uint8))]) as store:
        ids = np.arange(n, dtype=np.uint32)  # every row its own group
        pointer = store.upload_column(ids)
        with pytest.raises(SiloGpuError):
            store.group_count_hashed(None, [pointer, pointer], [n, n], 100)
        got_ids, got_counts = store.group_count_hashed(None, [pointer, pointer], [n, n], n)
        assert len(got_ids) == n and int(got_counts.sum()) == n
        store.free(pointer)


@pytest.mark.parametrize("n_symbols", [5, 22])
def test_mutations_select_threshold_arithmetic_matches_host_doubles(built, n_symbols):
    """K4: ceil((double)total * minProportion) - 1 on the device is the host's IEEE arithmetic bit for bit, for totals up
    to tens of millions and proportions that sit on rounding edges (mutations.cpp:197-211)."""
    rng = np.random.default_rng(n_symbols)
    positions = 20000
    counts = np.zeros((positions, n_symbols), dtype=np.uint32)
    scale = rng.choice([1, 10, 1000, 100000, 10_000_000], size=positions)
    for s in range(n_symbols):
        counts[:, s] = (rng.random(positions) ** 4 * scale).astype(np.uint32)
    counts[rng.random(positions) < 0.05] = 0                       # positions nobody covers
    reference = rng.integers(0, n_symbols, size=positions).astype(np.uint8)
    reference[rng.random(positions) < 0.02] = 0xFF                 # reference symbol not among the valid ones
    totals = counts.sum(axis=1, dtype=np.uint64)
    with make_store(64, [dict(name="s", alphabet="nuc", reference=np.ones(4, dtype=np.uint8))]) as store:
        for proportion in (0.0, 0.05, 1.0 / 3.0, 0.1, 0.3, 1e-9, 0.9999999, 1.0, 0.07, 2.0 / 7.0):
            if proportion == 0:
                threshold = np.zeros(positions, dtype=np.uint64)
            else:
                threshold = (np.ceil(totals.astype(np.float64) * proportion) - 1).astype(np.int64).astype(np.uint64) & 0xFFFFFFFF
            want = set()
            for position in np.nonzero(totals)[0]:
                for s in range(n_symbols):
                    if s != reference[position] and counts[position, s] > threshold[position]:
                        want.add((int(position), s, int(counts[position, s]), int(totals[position])))
            n, rows = store.mutations_select(counts, reference, proportion, positions * n_symbols)
            assert n == len(want), proportion
            assert {tuple(int(v) for v in row) for row in rows} == want, proportion
            # a list that is too short still reports the true number of selected cells
            n_short, rows_short = store.mutations_select(counts, reference, proportion, 7)
            assert n_short == n and len(rows_short) == min(n, 7)


# ---- the reference's operator known-answer vectors through k_filter_eval ------------------------------------------------
# tests/golden/operators/operator_vectors.json holds the sets of threshold.test.cpp:41-312, intersection.test.cpp:67-121,
# union.test.cpp:28-97, complement.test.cpp:10-47 and bitmap_selection.test.cpp:7-43 as data.  Each vector is lowered to a
# bit-program the way host/operators.cpp lowers the operator (n-ary leaf runs, the bit-sliced counter for Threshold) and
# run by silo_gpu_filter_eval; ids past row_count in a few expected sets are artefacts of the reference's flip and are
# clipped, as in tests/test_oracle_golden.py.
def _operator_vectors():
    import json
    import os

    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "operators", "operator_vectors.json")
    return json.load(open(path))


def _run_vector(row_count, leaf_sets, code, n_slots):
    """Evaluates `code` over leaves = the given id sets on a store of row_count rows; returns (sorted ids, count)."""
    ref = np.ones(2, dtype=np.uint8)
    with make_store(max(row_count, 1), [dict(name="s", alphabet="nuc", reference=ref)]) as store:
        leaves = []
        for ids in leaf_sets:
            mask = np.zeros(max(row_count, 1), dtype=bool)
            mask[[i for i in ids if i < row_count]] = True
            ptr = store.bitset_alloc()
            store.bitset_upload(ptr, dense.pack_bits(mask))
            leaves.append(ptr)
        words, count = _eval(store, code, leaves, n_slots)
        bits = dense.unpack_bits(words, store.row_words * 64)
        rows = max(row_count, 1)  # a vector over zero rows runs on a one-row store and is read clipped to no rows
        assert not bits[rows:].any()  # nothing outside [0, row_count): Complement is ~x & valid
        ids = [int(i) for i in np.nonzero(bits[:row_count])[0]]
        assert count == int(bits[:rows].sum())
        return ids


def test_reference_threshold_vectors_on_the_device(built):
    from silo_amd import binding as b

    for vec in _operator_vectors()["threshold"]:
        rc = vec["row_count"]
        positive, negative = vec["non_negated"], vec["negated"]
        k = len(positive) + len(negative)
        bits = max(1, k.bit_length())
        for case in vec["cases"]:
            code = []
            for bit in range(bits):
                code += b.encode(b.OP_ZERO, 1 + bit)
            if positive:
                code += b.encode(b.OP_CNT_ADD_N, 1, 0, bits, imm=0 | (len(positive) << 16))
            if negative:
                code += b.encode(b.OP_CNT_ADD_NOT_N, 1, 0, bits, imm=len(positive) | (len(negative) << 16))
            code += b.encode(b.OP_CNT_EQ if case["exact"] else b.OP_CNT_GE, 0, 1, bits, imm=case["n"])
            got = _run_vector(rc, positive + negative, code, 1 + bits)
            assert got == [i for i in case["expected"] if i < rc], (vec["cite"], case)
            # the same through single-leaf CNT_ADD instructions (composite children take this form)
            code = []
            for bit in range(bits):
                code += b.encode(b.OP_ZERO, 1 + bit)
            for leaf in range(len(positive)):
                code += b.encode(b.OP_CNT_ADD, 1, b.LEAF_OPERAND + leaf, bits)
            for leaf in range(len(positive), k):
                code += b.encode(b.OP_NOT, 0, b.LEAF_OPERAND + leaf) + b.encode(b.OP_CNT_ADD, 1, 0, bits)
            code += b.encode(b.OP_CNT_EQ if case["exact"] else b.OP_CNT_GE, 0, 1, bits, imm=case["n"])
            assert _run_vector(rc, positive + negative, code, 1 + bits) == got


def test_reference_intersection_union_complement_vectors_on_the_device(built):
    from silo_amd import binding as b

    vectors = _operator_vectors()
    for vec in vectors["intersection"]:
        positive, negative = vec["non_negated"], vec["negated"]
        code = (b.encode(b.OP_AND_N, 0, imm=0 | (len(positive) << 16)) if len(positive) > 1 else b.encode(b.OP_MOV, 0, b.LEAF_OPERAND))
        if len(negative) > 1:
            code += b.encode(b.OP_OR_N, 1, imm=len(positive) | (len(negative) << 16)) + b.encode(b.OP_ANDNOT, 0, 0, 1)
        elif negative:
            code += b.encode(b.OP_ANDNOT, 0, 0, b.LEAF_OPERAND + len(positive))
        assert _run_vector(vec["row_count"], positive + negative, code, 2) == vec["expected"], vec["cite"]
    for vec in vectors["union"]:
        children = vec["children"]
        if not children:
            code = b.encode(b.OP_ZERO, 0)
        elif len(children) == 1:
            code = b.encode(b.OP_MOV, 0, b.LEAF_OPERAND)
        else:
            code = b.encode(b.OP_OR_N, 0, imm=0 | (len(children) << 16))
        assert _run_vector(vec["row_count"], children, code, 1) == vec["expected"], vec["cite"]
    for vec in vectors["complement"]:
        assert _run_vector(vec["row_count"], [vec["child"]], b.encode(b.OP_NOT, 0, b.LEAF_OPERAND), 1) == vec["expected"], vec["cite"]
    for vec in vectors["leaf_operators"]:
        if vec["operator"] in ("Full", "Empty"):
            code = b.encode(b.OP_ONES if vec["operator"] == "Full" else b.OP_ZERO, 0)
            assert _run_vector(vec["row_count"], [], code, 1) == vec["expected"], vec["cite"]


def test_reference_bitmap_selection_vector_on_the_device(built):
    """bitmap_selection.test.cpp:8-26 probes ROW-wise bitmaps for a value (a position); the dense store keeps that as
    the missing symbol's column plane: CONTAINS is the plane, NOT_CONTAINS its complement (host/operators.cpp)."""
    from silo_amd import binding as b

    for vec in _operator_vectors()["bitmap_selection"]:
        rows = vec["rows"]
        column = [row for row, values in enumerate(rows) if vec["value"] in values]  # the transposed view the device holds
        assert _run_vector(len(rows), [column], b.encode(b.OP_MOV, 0, b.LEAF_OPERAND), 1) == vec["contains"]
        assert _run_vector(len(rows), [column], b.encode(b.OP_NOT, 0, b.LEAF_OPERAND), 1) == vec["not_contains"]


@pytest.mark.parametrize("n", [100, 70_000, 300_001])
def test_filter_eval_batch_matches_numpy_and_single_launches(built, n):
    """k_filter_eval_batch: many bit-programs of one store in one launch (programs with few and with many slots go in
    separate launches of the same call) — counts and bitsets equal to numpy and to silo_gpu_filter_eval one by one."""
    from silo_amd import binding as b

    rng = np.random.default_rng(n + 1)
    masks = [rng.random(n) < p for p in (0.5, 0.2, 0.7, 0.05, 0.9, 0.33, 0.01, 0.6)]
    ref = np.ones(4, dtype=np.uint8)
    with make_store(n, [dict(name="s", alphabet="nuc", reference=ref)]) as store:
        leaves = []
        for m in masks:
            ptr = store.bitset_alloc()
            store.bitset_upload(ptr, dense.pack_bits(m))
            leaves.append(ptr)
        L = b.LEAF_OPERAND
        total = sum(m.astype(int) for m in masks)
        programs, wants = [], []

        def add(code, n_slots, want, program_leaves=None):
            programs.append((code, leaves if program_leaves is None else program_leaves, n_slots))
            wants.append(want)

        add(b.encode(b.OP_OR_N, 0, imm=0 | (8 << 16)), 1, np.logical_or.reduce(masks))
        add(b.encode(b.OP_AND_N, 0, imm=1 | (3 << 16)), 1, masks[1] & masks[2] & masks[3])
        for k in range(0, 10):  # k-of-8 with a 4-bit counter in slots 1..4
            zero = b.encode(b.OP_ZERO, 1) + b.encode(b.OP_ZERO, 2) + b.encode(b.OP_ZERO, 3) + b.encode(b.OP_ZERO, 4)
            add(zero + b.encode(b.OP_CNT_ADD_N, 1, 0, 4, imm=0 | (8 << 16)) + b.encode(b.OP_CNT_GE, 0, 1, 4, imm=k), 5, total >= k)
            add(zero + b.encode(b.OP_CNT_ADD_NOT_N, 1, 0, 4, imm=0 | (8 << 16)) + b.encode(b.OP_CNT_EQ, 0, 1, 4, imm=k), 5, (8 - total) == k)
        add(b.encode(b.OP_AND, 0, L + 0, L + 1) + b.encode(b.OP_NOT, 1, L + 2) + b.encode(b.OP_OR, 0, 0, 1) + b.encode(b.OP_ANDNOT, 0, 0, L + 3), 2,
            ((masks[0] & masks[1]) | ~masks[2]) & ~masks[3])
        add(b.encode(b.OP_ONES, 0), 1, np.ones(n, bool))
        add(b.encode(b.OP_ZERO, 0), 1, np.zeros(n, bool))
        add(b.encode(b.OP_MOV, 0, L + 0), 1, masks[6], [leaves[6]])  # a program with its own (short) leaf list
        # slot-hungry programs (second and third launch class): the value travels through slots 11 and 30
        add(b.encode(b.OP_MOV, 11, L + 4) + b.encode(b.OP_AND, 0, 11, L + 5), 12, masks[4] & masks[5])
        add(b.encode(b.OP_MOV, 30, L + 7) + b.encode(b.OP_NOT, 31, 30) + b.encode(b.OP_OR, 0, 31, L + 1), 32, ~masks[7] | masks[1])
        outs = [store.bitset_alloc() for _ in programs]
        counts = store.filter_eval_batch(programs, outs)
        assert counts == [int(w.sum()) for w in wants]
        for out, want, (code, program_leaves, n_slots) in zip(outs, wants, programs):
            words = store.bitset_download(out)
            assert np.array_equal(dense.unpack_bits(words, n), want)
            assert not dense.unpack_bits(words, store.row_words * 64)[n:].any()
            single_words, single_count = _eval(store, code, program_leaves, n_slots)
            assert np.array_equal(single_words, words) and single_count == int(want.sum())
        # count only, and the empty batch
        assert store.filter_eval_batch(programs) == counts
        assert store.filter_eval_batch([]) == []
        # a bad operand is refused on the host, nothing is launched
        with pytest.raises(b.SiloGpuError):
            store.filter_eval_batch(programs + [(b.encode(b.OP_AND, 0, 5, 6), leaves, 2)])


@pytest.mark.parametrize("n,alphabet", [(140000, "nuc"), (70000, "aa"), (1000, "nuc")])
def test_two_pass_build_gives_the_same_store(built, n, alphabet):
    """silo_gpu_store_build_pass: the sequences streamed twice — counted, then written straight into the adaptive planes chosen
    from the counts, without build-time planes — give the store that finalize re-encodes out of build-time planes: the same
    layout, the same keys, the same answers.  Batches with unaligned boundaries (shared words), IUPAC codes and missing
    symbols; a store of short rows keeps its identity planes either way."""
    rng = np.random.default_rng(n + 23)
    positions = 61
    sym = skewed_symbols(rng, n, positions, alphabet)
    settle_positions(rng, sym, list(range(5, 30)) + list(range(40, 58)), alphabet)
    sym[:, 33] = rng.integers(0, 16 if alphabet == "nuc" else 25, size=n)  # a position that keeps its identity planes
    chars = NUC_CHARS if alphabet == "nuc" else AA_CHARS
    table_size = 16 if alphabet == "nuc" else 25
    cuts = [0, n // 3 + 17, 2 * n // 3 + 5, n]
    masks = [rng.random(n) < 0.4, np.ones(n, bool), rng.random(n) < 0.001]
    stores = []
    try:
        for two_pass in (False, True):
            store = make_store(n, [dict(name="a", alphabet=alphabet, reference=sym[0].copy())])
            stores.append(store)
            if two_pass:
                store.build_pass(0, 1)
                assert store.build_mode(0) == 1
                for a, b in zip(cuts[:-1], cuts[1:]):
                    store.append_sequences(0, a, chars[sym[a:b]])
                store.build_pass(0, 2)
                assert store.build_mode(0) == (2 if n >= 65536 else 0)
            for a, b in zip(cuts[:-1], cuts[1:]):
                store.append_sequences(0, a, chars[sym[a:b]])
            store.finalize()
            assert store.build_mode(0) == 0
        one, two = stores
        assert two.scan_rows(0, 0, positions) == one.scan_rows(0, 0, positions) and two.scan_escapes(0) == one.scan_escapes(0)
        assert [two.scan_rows(0, p, p + 1) for p in range(positions)] == [one.scan_rows(0, p, p + 1) for p in range(positions)]
        assert two.device_bytes == one.device_bytes
        scan_symbols = list(one.scan_symbols[0])
        for mask in masks:
            want = dense.mutation_counts(sym, mask, scan_symbols)
            for store in stores:
                ptr = store.bitset_alloc()
                store.bitset_upload(ptr, dense.pack_bits(mask))
                assert np.array_equal(store.mutations_scan(0, ptr), want)
        assert np.array_equal(two.mutations_scan(0, None), dense.mutation_counts(sym, masks[1], scan_symbols))  # the cached totals = the first pass's counts
        for position in (0, 7, 33, 50, positions - 1):
            for symbol in range(table_size):
                want_plane = dense.pack_bits(sym[:, position] == symbol)
                got = two.plane_download(0, position, symbol)
                assert np.array_equal(got[: len(want_plane)], want_plane) and not got[len(want_plane):].any(), (position, symbol)
        picked = rng.choice(n, size=50, replace=False).astype(np.uint32)
        assert np.array_equal(two.reconstruct_sequences(0, picked), chars[sym[picked]])
    finally:
        for store in stores:
            store.close()


@pytest.mark.parametrize("n,alphabet,positions,run_share", [(70000, "nuc", 397, 0.5), (6000, "aa", 4, 0.0005), (150001, "nuc", 512, 0.5)])
def test_missing_symbol_becomes_runs(built, n, alphabet, positions, run_share):
    """finalize() turns the plane of the missing symbol (N / X) into the sorted list of its runs and releases the plane: runs
    at the first and at the last position, rows that are missing throughout (null genomes), single cells, neighbouring runs;
    position counts that are and are not multiples of the kernel's look-ahead.  Every position's plane (filter leaves),
    FastaAligned and the scan answer as with the plane kept (SILO_GPU_TUNE_MISSING_RUNS < 0)."""
    rng = np.random.default_rng(n + positions)
    sym = random_symbols(rng, n, positions, alphabet)
    missing = 15 if alphabet == "nuc" else 24
    chars = NUC_CHARS if alphabet == "nuc" else AA_CHARS
    sym[sym == missing] = 1  # the missing cells are placed below
    start = rng.integers(0, positions, size=n)
    length = rng.geometric(0.2, size=n)
    has_run = rng.random(n) < run_share
    columns = np.arange(positions)[None, :]
    in_run = has_run[:, None] & (columns >= start[:, None]) & (columns < (start + length)[:, None])
    second = (has_run & (rng.random(n) < 0.2))[:, None] & (columns >= (start + length + 1)[:, None]) & (columns < (start + length + 3)[:, None])
    sym[in_run | second] = missing
    sym[rng.random((n, positions)) < 0.001] = missing      # single cells
    sym[rng.choice(n, size=7, replace=False)] = missing     # missing throughout
    sym[:3, 0] = missing
    sym[3:6, positions - 1] = missing
    mask = rng.random(n) < 0.5
    answers = {}
    for knob in (0, -1, "two passes"):  # runs out of the plane at finalize; the plane kept; runs written while the rows stream in
        with make_store(n, [dict(name="a", alphabet=alphabet, reference=sym[0].copy())]) as store:
            store.tune(8, knob if knob == -1 else 0)
            store.tune(9, -1)  # (no charge per kind of launch: with it these short, uniformly random rows would all become one-hot rows + keys)
            try:
                if knob == "two passes":
                    store.build_pass(0, 1)
                    store.append_sequences(0, 0, chars[sym[: n // 2 + 3]])
                    store.append_sequences(0, n // 2 + 3, chars[sym[n // 2 + 3:]])
                    store.build_pass(0, 2)
                store.append_sequences(0, 0, chars[sym[: n // 2 + 3]])
                store.append_sequences(0, n // 2 + 3, chars[sym[n // 2 + 3:]])
                store.finalize()
            finally:
                store.tune(8, 0)
                store.tune(9, 0)
            if knob != "two passes" or n >= 65536:
                assert (store.plane(0, 0, missing) is None) == (knob != -1)  # no resident plane once it has become runs
            for position in sorted(set(range(0, positions, 7)) | {1, positions - 2, positions - 1}):
                got = store.plane_download(0, position, missing)
                want = dense.pack_bits(sym[:, position] == missing)
                assert np.array_equal(got[: len(want)], want) and not got[len(want):].any(), (knob, position)
            picked = np.concatenate([np.arange(8), rng.choice(n, size=60, replace=False), np.nonzero((sym == missing).all(axis=1))[0]]).astype(np.uint32)
            assert np.array_equal(store.reconstruct_sequences(0, picked), chars[sym[picked]])
            ptr = store.bitset_alloc()
            store.bitset_upload(ptr, dense.pack_bits(mask))
            assert np.array_equal(store.mutations_scan(0, ptr), dense.mutation_counts(sym, mask, list(store.scan_symbols[0])))
            answers[knob] = store.device_bytes
    assert answers[0] < answers[-1] and answers["two passes"] < answers[-1]


def test_append_from_unaligned_pageable_memory_in_back_to_back_batches(built):
    """Regression test for the path behind round 1's GPU memory-access fault: silo_gpu_store_append_sequences
    fed from PAGEABLE host memory at an address that is not even 2-byte aligned, rows of a length that is not a multiple
    of 4 (so a row's last characters share their 4-byte word with the next row), several batches back to back with
    boundaries inside a 64-sequence word, null genomes in between — every plane equal to the naive transposition."""
    rng = np.random.default_rng(99)
    n, positions = 9_001, 1_003
    sym = random_symbols(rng, n, positions, "nuc")
    is_null = (rng.random(n) < 0.03).astype(np.uint8)
    effective = sym.copy()
    effective[is_null.astype(bool)] = 15
    ref = rng.integers(1, 5, size=positions).astype(np.uint8)
    with make_store(n, [dict(name="main", alphabet="nuc", reference=ref)]) as store:
        first = 0
        for size in (1, 63, 1, 2_000, 129, 4_096, 7, 2_704):  # sums to n; none of the boundaries is word aligned
            backing = np.empty(size * positions + 3, dtype=np.uint8)       # pageable numpy memory
            chars = backing[3:3 + size * positions].reshape(size, positions)  # ... entered 3 bytes into it
            chars[:] = NUC_CHARS[sym[first:first + size]]
            assert chars.ctypes.data % 2 == 1 or chars.ctypes.data % 4 != 0
            store.append_sequences(0, first, chars, is_null[first:first + size])
            first += size
        assert first == n
        store.finalize()
        for p in (0, 1, 500, positions - 2, positions - 1):
            for s in range(16):
                want = dense.pack_bits(effective[:, p] == s)
                got = store.plane_download(0, p, s)
                assert np.array_equal(got[: len(want)], want), (p, s)
                assert not got[len(want):].any()
        counts = store.mutations_scan(0, None)
        assert np.array_equal(counts, dense.mutation_counts(effective, np.ones(n, bool), list(store.scan_symbols[0])))
