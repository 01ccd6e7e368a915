"""CPU twin of k_generate_synthetic (lapis-silo_amd/csrc/silo_gpu.hip) — test infrastructure only.

Written from the model specification in DESIGN.md §6, independently of the device code: given the
same SynthModel parameters it must reproduce symbol(sequence, position) bit-exactly (all arithmetic
is wrapping 64-bit integer).  numpy, vectorised over a (sequences x positions) block.
"""
import numpy as np

M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def mix64(z):
    """splitmix64 finaliser on a uint64 array."""
    with np.errstate(over="ignore"):
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return z ^ (z >> np.uint64(31))


def symbol_matrix(model, sequence_ids, positions):
    """uint8 [len(sequence_ids)][len(positions)] symbol ids of the synthetic alignment."""
    seq = np.asarray(sequence_ids, dtype=np.uint64)[:, None]
    pos = np.asarray(positions, dtype=np.uint64)[None, :]
    sidx = np.asarray(sequence_ids, dtype=np.int64)
    pidx = np.asarray(positions, dtype=np.int64)
    n_positions = np.uint64(model.positions)
    nuc = model.alphabet == "nuc"
    missing = 15 if nuc else 24
    private_base, private_count = (1, 4) if nuc else (1, 20)
    ambiguous_base, ambiguous_count = (5, 10) if nuc else (21, 2)

    with np.errstate(over="ignore"):
        seq_hash = np.uint64(model.seed) ^ (seq * np.uint64(0x9E3779B97F4A7C15))
        h = mix64(seq_hash ^ (pos * np.uint64(0xC2B2AE3D27D4EB4F)))

    lineage = model.lineage_of_sequence[sidx].astype(np.int64)
    table = model.lineage_symbol[pidx][:, lineage].T  # [seq][pos]
    reference = np.broadcast_to(model.reference[pidx][None, :], table.shape)
    out = np.where(table != 0xFF, table, reference).astype(np.uint8)

    ambiguous = ((h >> np.uint64(32)) & np.uint64(0xFFFFFF)) < np.uint64(model.ambiguous_threshold)
    amb_symbol = (ambiguous_base + ((h >> np.uint64(56)) % np.uint64(ambiguous_count))).astype(np.uint8)
    out = np.where(ambiguous, amb_symbol, out)

    private = (h & np.uint64(0xFFFFF)) < np.uint64(model.private_threshold)
    priv_symbol = (private_base + (((h >> np.uint64(20)) & np.uint64(0xFFF)) % np.uint64(private_count))).astype(np.uint8)
    out = np.where(private, priv_symbol, out)

    mstart = model.missing_start[sidx].astype(np.uint64)[:, None]
    mlen = model.missing_len[sidx].astype(np.uint64)[:, None]
    in_missing = (pos >= mstart) & (pos < mstart + mlen)
    out = np.where(in_missing, np.uint8(missing), out)

    lead = model.lead_gap[sidx].astype(np.uint64)[:, None]
    trail = model.trail_gap[sidx].astype(np.uint64)[:, None]
    in_gap = (pos < lead) | (pos >= n_positions - trail)
    out = np.where(in_gap, np.uint8(0), out)
    return out.astype(np.uint8)
