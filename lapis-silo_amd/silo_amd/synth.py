"""Host-side parameters of the synthetic SARS-CoV-2-shaped data set.

Only *parameters* are made here (per-sequence lineage / gap runs / missing run, and the per-lineage
table of defining substitutions); the cells themselves are a pure integer function
symbol(seed, sequence, position) evaluated on the GPU by k_generate_synthetic and, independently, on
the CPU by oracle/synth.py — so any slice of a 10 M-sequence store can be reproduced without ever
holding it on the host.
"""
from dataclasses import dataclass
from typing import List, Optional

import numpy as np

from . import alphabet as alpha

DEFAULT_SEED = 0x5110C0DE
SYMBOL_NONE = 0xFF


@dataclass
class LineageTree:
    names: List[str]
    parent: np.ndarray  # int32 [L], -1 for the root
    weights: np.ndarray  # float64 [L], Zipf(s=1.1)

    @property
    def n_lineages(self):
        return len(self.names)

    def subtree(self, root: int) -> np.ndarray:
        """uint8 membership [L]: root and all of its descendants (the sublineage set)."""
        member = np.zeros(self.n_lineages, dtype=np.uint8)
        member[root] = 1
        for k in range(root + 1, self.n_lineages):  # parents precede children
            if self.parent[k] >= 0 and member[self.parent[k]]:
                member[k] = 1
        return member


def make_lineage_tree(n_lineages: int, fanout: int = 3) -> LineageTree:
    parent = np.full(n_lineages, -1, dtype=np.int32)
    names = ["B"]
    for k in range(1, n_lineages):
        parent[k] = (k - 1) // fanout
        names.append(f"{names[parent[k]]}.{(k - 1) % fanout + 1}")
    weights = 1.0 / np.power(np.arange(1, n_lineages + 1, dtype=np.float64), 1.1)
    return LineageTree(names, parent, weights / weights.sum())


@dataclass
class SynthModel:
    seed: int
    alphabet: str
    reference: np.ndarray           # uint8 [P]
    n_lineages: int
    lineage_of_sequence: np.ndarray  # uint16 [N]
    lead_gap: np.ndarray            # uint32 [N]
    trail_gap: np.ndarray
    missing_start: np.ndarray
    missing_len: np.ndarray
    lineage_symbol: np.ndarray      # uint8 [P][L], 0xFF = reference
    private_threshold: int          # of 2**20
    ambiguous_threshold: int        # of 2**24

    @property
    def n_sequences(self):
        return len(self.lineage_of_sequence)

    @property
    def positions(self):
        return len(self.reference)


def assign_lineages(n_sequences: int, tree: LineageTree, seed: int) -> np.ndarray:
    rng = np.random.Generator(np.random.PCG64(seed))
    cumulative = np.cumsum(tree.weights)
    cumulative[-1] = 1.0
    u = rng.random(n_sequences)
    return np.searchsorted(cumulative, u, side="right").astype(np.uint16)


def make_lineage_table(reference: np.ndarray, alphabet: str, tree: LineageTree, seed: int) -> np.ndarray:
    """[P][L] uint8: symbol a lineage carries at a position, 0xFF where it has the reference symbol.
    A lineage inherits its parent's substitutions and adds 2-8 of its own (root: 10)."""
    rng = np.random.Generator(np.random.PCG64(seed ^ 0xA5A5A5A5))
    positions = len(reference)
    n = tree.n_lineages
    a = alpha.ALPHABETS[alphabet]
    substitutes = [1, 2, 3, 4] if alphabet == "nuc" else list(range(1, 21))
    table = np.full((n, positions), SYMBOL_NONE, dtype=np.uint8)
    for k in range(n):
        if tree.parent[k] >= 0:
            table[k] = table[tree.parent[k]]
        n_new = 10 if k == 0 else int(rng.integers(2, 9))
        n_new = min(n_new, positions)
        for p in rng.choice(positions, size=n_new, replace=False):
            if rng.random() < 0.05:
                symbol = 0  # deletion
            else:
                symbol = int(substitutes[int(rng.integers(len(substitutes)))])
            table[k, p] = SYMBOL_NONE if symbol == reference[p] else symbol
    assert a.missing not in substitutes
    return np.ascontiguousarray(table.T)


def make_model(
    n_sequences: int,
    reference: np.ndarray,
    alphabet: str,
    tree: LineageTree,
    lineage_of_sequence: np.ndarray,
    seed: int = DEFAULT_SEED,
    store_index: int = 0,
    table_seed: Optional[int] = None,
) -> SynthModel:
    """Per-store model; `lineage_of_sequence` is shared by all sequence stores of a database.  table_seed: the seed of the
    table of lineage substitutions where it is not `seed` — the sequence-id shards of one database share the lineages (and
    so the table) and differ in their rows (`seed`)."""
    reference = np.ascontiguousarray(reference, dtype=np.uint8)
    positions = len(reference)
    store_seed = (seed + 0x9E3779B9 * (store_index + 1)) & 0xFFFFFFFFFFFFFFFF
    rng = np.random.Generator(np.random.PCG64(store_seed))
    nuc = alphabet == "nuc"
    gap_prob, lead_mean, trail_mean = (0.99, 54.0, 66.0) if nuc else (0.02, 3.0, 3.0)
    missing_prob, missing_mean = (0.5, 300.0) if nuc else (0.1, 30.0)

    def runs(prob, mean, limit):
        present = rng.random(n_sequences) < prob
        length = rng.geometric(1.0 / mean, size=n_sequences)
        return np.where(present, np.minimum(length, limit), 0).astype(np.uint32)

    lead = runs(gap_prob, lead_mean, max(1, positions // 4))
    trail = runs(gap_prob, trail_mean, max(1, positions // 4))
    missing_len = runs(missing_prob, missing_mean, max(1, positions // 2))
    missing_start = (rng.random(n_sequences) * (positions - missing_len)).astype(np.uint32)

    private_threshold = max(1, int(round(3.0 / 29903.0 * (1 << 20)))) if nuc else max(1, int(round((1 << 20) / 9814.0)))
    ambiguous_threshold = int(round(1e-5 * (1 << 24)))
    return SynthModel(
        seed=store_seed,
        alphabet=alphabet,
        reference=reference,
        n_lineages=tree.n_lineages,
        lineage_of_sequence=np.ascontiguousarray(lineage_of_sequence, dtype=np.uint16),
        lead_gap=lead,
        trail_gap=trail,
        missing_start=missing_start,
        missing_len=missing_len,
        lineage_symbol=make_lineage_table(
            reference, alphabet, tree, store_seed if table_seed is None else (table_seed + 0x9E3779B9 * (store_index + 1)) & 0xFFFFFFFFFFFFFFFF
        ),
        private_threshold=private_threshold,
        ambiguous_threshold=ambiguous_threshold,
    )


def random_reference(positions: int, alphabet: str, seed: int) -> np.ndarray:
    rng = np.random.Generator(np.random.PCG64(seed))
    if alphabet == "nuc":
        return rng.integers(1, 5, size=positions).astype(np.uint8)
    ref = rng.integers(1, 21, size=positions).astype(np.uint8)
    ref[-1] = 23  # genes end in a stop codon
    return ref
