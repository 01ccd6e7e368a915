"""The roaring-format CPU port (bench cpu_baseline) against the naive checker and the synthetic twin."""
import numpy as np
import pytest

from oracle import cpu_port, dense
from oracle import synth as oracle_synth


def skewed_symbols(rng, n, positions, alphabet):
    if alphabet == "nuc":
        probs = np.array([0.02, 0.6, 0.1, 0.1, 0.1] + [0.001] * 10 + [0.07])
    else:
        probs = np.array([0.02] + [0.04] * 20 + [0.005, 0.005, 0.02, 0.15])
    probs = probs / probs.sum()
    sym = rng.choice(len(probs), size=(n, positions), p=probs).astype(np.uint8)
    sym[:, 0] = 15 if alphabet == "nuc" else 24   # a position where every row is missing: nothing is deleted
    sym[n // 3 : n // 2, 1] = 0                   # a long run -> run containers
    return sym


@pytest.mark.parametrize("n,alphabet", [(1, "nuc"), (70, "aa"), (5000, "nuc"), (140000, "nuc"), (70001, "aa")])
def test_port_scan_matches_naive(n, alphabet):
    rng = np.random.default_rng(n)
    positions = 9
    sym = skewed_symbols(rng, n, positions, alphabet)
    store = cpu_port.PortStore(n, 0, positions, alphabet, symbols=sym)
    n_symbols = 16 if alphabet == "nuc" else 25
    for density in (0.0, 0.003, 0.4, 1.0):
        mask = rng.random(n) < density
        filt = cpu_port.Filter(dense.pack_bits(mask), n)
        assert filt.cardinality == int(mask.sum())
        counts, _ = store.mutations_scan(filt, n_threads=2, grain=4)
        want = dense.mutation_counts(sym, mask, list(range(n_symbols)))
        missing = 15 if alphabet == "nuc" else 24
        want[:, missing] = 0  # the missing symbol is never indexed (sequence_store.cpp:124-128)
        assert np.array_equal(counts, want), density
    counts, _ = store.mutations_scan(None, n_threads=2)
    want = dense.mutation_counts(sym, np.ones(n, bool), list(range(n_symbols)))
    want[:, 15 if alphabet == "nuc" else 24] = 0
    assert np.array_equal(counts, want)
    store.close()


def test_containers_and_cardinality_and_contains():
    rng = np.random.default_rng(3)
    n = 300000
    masks = [rng.random(n) < 0.001, rng.random(n) < 0.3, np.zeros(n, bool), np.zeros(n, bool)]
    masks[2][1000:150000] = True
    masks[3][::2] = True
    filters = [cpu_port.Filter(dense.pack_bits(m), n) for m in masks]
    for a, ma in zip(filters, masks):
        for b, mb in zip(filters, masks):
            assert a.and_cardinality(b) == int((ma & mb).sum())
        for value in rng.integers(0, n, size=200):
            assert a.contains(int(value)) == bool(ma[value])


def test_synthetic_twin_in_c_matches_numpy_twin():
    import sys, os
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "lapis-silo_amd"))
    from silo_amd import synth

    n, positions = 4000, 150
    tree = synth.make_lineage_tree(30)
    lineage = synth.assign_lineages(n, tree, 5)
    for k, alphabet in enumerate(["nuc", "aa"]):
        ref = synth.random_reference(positions, alphabet, 17 + k)
        model = synth.make_model(n, ref, alphabet, tree, lineage, seed=99, store_index=k)
        model.ambiguous_threshold = 1 << 13
        begin, count = 20, 100
        sym = oracle_synth.symbol_matrix(model, np.arange(n), np.arange(begin, begin + count))
        store = cpu_port.PortStore(n, begin, count, alphabet, model=model)
        mask = tree.subtree(1)[lineage].astype(bool)
        counts, _ = store.mutations_scan(cpu_port.Filter(dense.pack_bits(mask), n), n_threads=2)
        n_symbols = 16 if alphabet == "nuc" else 25
        want = dense.mutation_counts(sym, mask, list(range(n_symbols)))
        want[:, 15 if alphabet == "nuc" else 24] = 0
        assert np.array_equal(counts, want)
        store.close()
