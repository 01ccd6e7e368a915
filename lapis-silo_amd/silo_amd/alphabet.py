"""Symbol alphabets, ids = the reference's enum values.

Nucleotide: include/silo/common/nucleotide_symbols.h:15-84, src/silo/common/nucleotide_symbols.cpp:46-85
AminoAcid:  include/silo/common/aa_symbols.h:15-89,        src/silo/common/aa_symbols.cpp:62-117
"""
from dataclasses import dataclass, field
from typing import Dict, List


@dataclass(frozen=True)
class Alphabet:
    name: str
    abi_id: int                      # SILO_GPU_ALPHABET_*
    chars: str                       # index = enum value
    symbols: List[int]               # SYMBOLS iteration order
    valid_mutation_symbols: List[int]
    missing: int                     # SYMBOL_MISSING
    char_to_symbol: Dict[str, int] = field(default_factory=dict)

    def symbol_to_char(self, symbol: int) -> str:
        return self.chars[symbol]

    @property
    def count(self) -> int:
        return len(self.chars)


def _nucleotide():
    chars = "-ACGTRYSWKMBDHVN"
    table = {c: i for i, c in enumerate(chars)}
    table["."] = 0  # nucleotide_symbols.cpp:48-50
    table["U"] = 4  # nucleotide_symbols.cpp:58-60
    return Alphabet("nuc", 0, chars, list(range(16)), [0, 1, 2, 3, 4], 15, table)


def _amino_acid():
    # enum order: ... Y(20) B(21) Z(22) STOP(23) X(24)        aa_symbols.h:15-41
    chars = "-ACDEFGHIKLMNPQRSTVWYBZ*X"
    table = {c: i for i, c in enumerate(chars)}
    # SYMBOLS iteration order puts X before STOP              aa_symbols.h:49-54
    symbols = list(range(23)) + [24, 23]
    valid = list(range(21)) + [23]                           # aa_symbols.h:56-79
    return Alphabet("aa", 1, chars, symbols, valid, 24, table)


NUCLEOTIDE = _nucleotide()
AMINO_ACID = _amino_acid()
ALPHABETS = {"nuc": NUCLEOTIDE, "aa": AMINO_ACID}
