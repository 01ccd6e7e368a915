"""CPU restatement of the reference's filter -> Aggregated / Mutations path.  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module; the
product (lapis-silo_amd/) never does.

What is restated (file:line under the reference tree, pflanze/LAPIS-SILO @ 2025-01-17):
  alphabets            src/silo/common/nucleotide_symbols.cpp:46-85, aa_symbols.cpp:62-117,
                       include/silo/common/{nucleotide_symbols,aa_symbols}.h
  storage              src/silo/storage/position.cpp:24-39,42-127 (flipped / deleted bitmaps),
                       src/silo/storage/sequence_store.cpp:100-220 (fillIndexes, fillNBitmaps,
                       optimizeBitmaps), src/silo/storage/column/pango_lineage_column.cpp:21-77,
                       src/silo/storage/pango_lineage_alias.cpp, src/silo/common/pango_lineage.cpp:25-35
  filter expressions   src/silo/query_engine/filter_expressions/{expression,and,or,negation,nof,maybe,
                       exact,true,false,nucleotide_symbol_equals,aa_symbol_equals,has_mutation,
                       has_aa_mutation,pango_lineage_filter}.cpp
  operators            src/silo/query_engine/operators/{index_scan,complement,intersection,union,
                       threshold,bitmap_selection,full,empty}.cpp
  actions              src/silo/query_engine/actions/{action,aggregated,mutations}.cpp
  engine               src/silo/query_engine/{query,query_engine,query_result}.cpp

The bitmap arithmetic itself lives in CRoaring 1.0.0 (conanfile.py:16), which is not vendored in the
reference tree; only its set semantics matter here (and_cardinality, andnot_cardinality, |, &, -,
flip(range), contains, iteration), so a bitmap is a Python int used as a bitset (bit i = row i).

PINNING: this restatement is pinned by the reference's own fixtures only — the end-to-end goldens of
endToEndTests/test/queries (tests/golden/queries) on testBaseData/exampleDataset and the operator unit
test vectors (tests/golden/operators).  The reference cannot be compiled here (CRoaring, oneTBB,
Boost, spdlog and a recent nlohmann_json are absent).
"""
import json
import math

import numpy as np
from dataclasses import dataclass, field
from typing import Dict, List, Optional

UINT32 = 0xFFFFFFFF


class QueryParseException(Exception):
    """query_parse_exception.h:13-16 -> HTTP 400 {"error": "Bad request"}."""


class QueryCompilationException(Exception):
    """query_compilation_exception.cpp -> HTTP 500."""


def check_silo_query(condition, message):
    if not condition:
        raise QueryParseException(message)


# ------------------------------------------------------------------------------------------------
# alphabets
# ------------------------------------------------------------------------------------------------
class Nucleotide:
    NAME_LOWER = "nucleotide"
    CHARS = "-ACGTRYSWKMBDHVN"  # enum order, nucleotide_symbols.h:15-32
    COUNT = 16
    SYMBOLS = list(range(16))
    VALID_MUTATION_SYMBOLS = [0, 1, 2, 3, 4]
    SYMBOL_MISSING = 15

    @staticmethod
    def char_to_symbol(char):
        if char in (".", "-"):
            return 0
        if char == "U":
            return 4
        idx = Nucleotide.CHARS.find(char)
        return idx if idx > 0 and len(char) == 1 else None

    @staticmethod
    def symbol_to_char(symbol):
        return Nucleotide.CHARS[symbol]


class AminoAcid:
    NAME_LOWER = "amino acid"
    CHARS = "-ACDEFGHIKLMNPQRSTVWYBZ*X"  # enum order: ... B Z STOP X, aa_symbols.h:15-41
    COUNT = 25
    GAP, B, Z, STOP, X = 0, 21, 22, 23, 24
    SYMBOLS = list(range(23)) + [24, 23]  # iteration order ... B Z X STOP, aa_symbols.h:49-54
    VALID_MUTATION_SYMBOLS = list(range(21)) + [23]  # aa_symbols.h:56-79
    SYMBOL_MISSING = 24

    @staticmethod
    def char_to_symbol(char):
        idx = AminoAcid.CHARS.find(char)
        return idx if idx >= 0 and len(char) == 1 else None

    @staticmethod
    def symbol_to_char(symbol):
        return AminoAcid.CHARS[symbol]


# nucleotide_symbol_equals.cpp:28-73
AMBIGUITY_NUC_SYMBOLS = {
    0: [0],
    1: [1, 5, 10, 8, 12, 13, 14, 15],   # A R M W D H V N
    2: [2, 6, 10, 7, 11, 13, 14, 15],   # C Y M S B H V N
    3: [3, 5, 9, 7, 11, 12, 14, 15],    # G R K S B D V N
    4: [4, 6, 9, 8, 11, 12, 13, 15],    # T Y K W B D H N
}
for _s in range(5, 16):
    AMBIGUITY_NUC_SYMBOLS[_s] = [_s]


def std_remove(values, value):
    """std::remove WITHOUT erase (nucleotide_symbol_equals.cpp:164-167, has_mutation.cpp:58-65,
    has_aa_mutation.cpp:48-52): kept elements are shifted to the front, the tail keeps its old values."""
    values = list(values)
    write = 0
    for item in values:
        if item != value:
            values[write] = item
            write += 1
    return values


# ------------------------------------------------------------------------------------------------
# bitsets
# ------------------------------------------------------------------------------------------------
def bits_from_ids(ids):
    out = 0
    for i in ids:
        out |= 1 << i
    return out


def ids_from_bits(bits):
    out = []
    i = 0
    while bits:
        low = bits & -bits
        i = low.bit_length() - 1
        out.append(i)
        bits ^= low
    return out


def _bits_from_mask(mask):
    """bool array -> int bitset (bit i = mask[i])."""
    return int.from_bytes(np.packbits(mask, bitorder="little").tobytes(), "little")


def card(bits):
    return bin(bits).count("1") if not hasattr(int, "bit_count") else bits.bit_count()


def flip(bits, begin, end):
    """roaring flip(begin, end)."""
    if end <= begin:
        return bits
    return bits ^ (((1 << (end - begin)) - 1) << begin)


# ------------------------------------------------------------------------------------------------
# storage
# ------------------------------------------------------------------------------------------------
class Position:
    """position.h:23-66 — per-symbol bitmaps with one optionally flipped / deleted symbol."""

    def __init__(self, alphabet):
        self.alphabet = alphabet
        self.bitmaps = {s: 0 for s in alphabet.SYMBOLS}
        self.flipped = None
        self.deleted = None

    @staticmethod
    def from_initially_flipped(alphabet, symbol):
        position = Position(alphabet)
        position.flipped = symbol
        return position

    def add_values(self, symbol, ids, current_offset, interval_size):  # position.cpp:24-39
        if symbol == self.deleted:
            return
        if ids:
            self.bitmaps[symbol] |= bits_from_ids(ids)
        if symbol == self.flipped:
            self.bitmaps[symbol] = flip(self.bitmaps[symbol], current_offset, current_offset + interval_size)

    def highest_cardinality_symbol(self, sequence_count):  # position.cpp:42-68
        if self.deleted is not None:
            raise RuntimeError("symbol currently deleted")
        max_symbol, max_count = None, 0
        for symbol in self.alphabet.SYMBOLS:
            count = card(self.bitmaps[symbol])
            if symbol == self.flipped:
                count = sequence_count - count
            if count > max_count:
                max_symbol, max_count = symbol, count
        return max_symbol

    def flip_most_numerous(self, sequence_count):  # position.cpp:70-99
        if self.deleted is not None:
            raise RuntimeError("symbol currently deleted")
        max_symbol = self.highest_cardinality_symbol(sequence_count)
        if max_symbol != self.flipped:
            if self.flipped is not None:
                self.bitmaps[self.flipped] = flip(self.bitmaps[self.flipped], 0, sequence_count)
            if max_symbol is not None:
                self.bitmaps[max_symbol] = flip(self.bitmaps[max_symbol], 0, sequence_count)
            self.flipped = max_symbol
            return self.flipped
        return None

    def delete_most_numerous(self, sequence_count):  # position.cpp:102-127
        if self.deleted is not None:
            raise RuntimeError("symbol currently deleted")
        if self.flipped is not None:
            self.bitmaps[self.flipped] = flip(self.bitmaps[self.flipped], 0, sequence_count)
            self.flipped = None
        max_symbol = self.highest_cardinality_symbol(sequence_count)
        if max_symbol is not None:
            self.bitmaps[max_symbol] = 0
            self.deleted = max_symbol
            return self.deleted
        return None


class SequenceStorePartition:
    """sequence_store.h:34-88."""

    BUFFER_SIZE = 1024  # sequence_store.cpp:34

    def __init__(self, alphabet, reference_sequence):
        self.alphabet = alphabet
        self.reference_sequence = list(reference_sequence)
        self.positions = [Position.from_initially_flipped(alphabet, s) for s in reference_sequence]
        self.missing_symbol_bitmaps: List[set] = []  # row-wise: positions where the row is N / X
        self.sequence_count = 0
        self.indexing_differences = []

    def fill(self, genomes):
        """genomes: list of str or None (sequence_store.cpp:31-66)."""
        for begin in range(0, len(genomes), self.BUFFER_SIZE):
            self.interpret(genomes[begin : begin + self.BUFFER_SIZE])
        self.optimize_bitmaps()

    def interpret(self, genomes):  # sequence_store.cpp:213-220
        """fillIndexes (:100-151) + fillNBitmaps (:153-190); numpy only vectorises the per-position
        id lists, the data flow (ids per symbol per position -> addValues) is the reference's."""
        length = len(self.positions)
        number = len(genomes)
        present = [i for i, g in enumerate(genomes) if g is not None]
        symbols = None
        if present:
            for i in present:
                if len(genomes[i]) != length:
                    raise ValueError("sequence length differs from the reference")
            lut = np.full(256, 255, dtype=np.uint8)
            for code in range(256):
                symbol = self.alphabet.char_to_symbol(chr(code))
                if symbol is not None:
                    lut[code] = symbol
            raw = np.frombuffer("".join(genomes[i] for i in present).encode("latin-1"), dtype=np.uint8)
            symbols = lut[raw.reshape(len(present), length)]
            if (symbols == 255).any():
                bad = raw.reshape(len(present), length)[symbols == 255][0]
                raise ValueError(f"Illegal character {int(bad)} contained in sequence.")
        present_rows = np.asarray(present, dtype=np.int64)
        missing = self.alphabet.SYMBOL_MISSING
        for p in range(length):
            position = self.positions[p]
            ids_per_symbol = {}
            if symbols is not None:
                column = symbols[:, p]
                for symbol in np.unique(column):
                    if symbol != missing:  # the missing symbol is never indexed (:124-128)
                        mask = np.zeros(number, dtype=bool)
                        mask[present_rows[column == symbol]] = True
                        ids_per_symbol[int(symbol)] = mask
            for symbol in self.alphabet.SYMBOLS:  # addSymbolsToPositions (:137-151) -> Position::addValues
                mask = ids_per_symbol.get(symbol)
                if symbol == position.deleted:
                    continue
                if mask is not None:
                    position.bitmaps[symbol] |= _bits_from_mask(mask) << self.sequence_count
                if symbol == position.flipped:
                    position.bitmaps[symbol] = flip(position.bitmaps[symbol], self.sequence_count, self.sequence_count + number)
        row_of = {i: k for k, i in enumerate(present)}
        for i in range(number):
            if genomes[i] is None:
                self.missing_symbol_bitmaps.append(set(range(length)))  # :166-169
            else:
                self.missing_symbol_bitmaps.append(set(np.nonzero(symbols[row_of[i]] == missing)[0].tolist()))
        self.sequence_count += number

    def optimize_bitmaps(self):  # sequence_store.cpp:192-211
        for p, position in enumerate(self.positions):
            changed = position.delete_most_numerous(self.sequence_count)
            if changed is not None:
                self.indexing_differences.append((p, changed))

    def get_bitmap(self, position, symbol):
        return self.positions[position].bitmaps[symbol]

    def symbol_at(self, position, row):
        """One cell of reconstructSequence (fasta_aligned.cpp:44-83): the reference symbol, overridden by the deleted
        (most numerous) symbol of the position, by any stored bitmap that holds the row, then by the missing symbol."""
        if getattr(self, "_differences_at", None) is None or len(self._differences_at) != len(self.indexing_differences):
            self._differences_at = dict(self.indexing_differences)  # position -> deleted symbol (one entry per position)
        symbol = self._differences_at.get(position, self.reference_sequence[position])
        entry = self.positions[position]
        for candidate in self.alphabet.SYMBOLS:
            if candidate != entry.flipped and candidate != entry.deleted and (entry.bitmaps[candidate] >> row) & 1:
                symbol = candidate
        if position in self.missing_symbol_bitmaps[row]:
            symbol = self.alphabet.SYMBOL_MISSING
        return symbol


def get_parent_lineages(value):  # pango_lineage.cpp:25-35
    out = []
    pos = 0
    while pos != -1:
        pos = value.find(".", pos + 1)
        out.append(value if pos == -1 else value[:pos])
    return out


class PangoLineageAliasLookup:  # pango_lineage_alias.cpp
    def __init__(self, alias_key):
        self.alias_key = {}
        for key, value in alias_key.items():  # readFromJson :88-102
            if isinstance(value, list):
                self.alias_key[key] = list(value)
            elif isinstance(value, str) and value:
                self.alias_key[key] = [value]

    def unalias(self, lineage):  # :21-41
        prefix, dot, suffix = lineage.partition(".")
        if prefix in self.alias_key:
            values = self.alias_key[prefix]
            if len(values) != 1:
                return lineage
            if not dot:
                return values[0]
            return values[0] + "." + "".join(suffix.split())
        return lineage


    def alias(self, unaliased):  # aliasPangoLineage :43-73: the longest proper prefix of >= 3 elements that is an alias value
        elements = unaliased.split(".")
        for i in range(len(elements), 3, -1):
            search_value = ".".join(elements[: i - 1])
            for alias, values in self.alias_key.items():
                if len(values) != 1:
                    continue
                if values[0] == search_value:
                    leftover = ".".join(elements[i - 1 :])
                    return alias + ("." + leftover if leftover else "")
        return unaliased


class PangoLineageColumnPartition:  # pango_lineage_column.cpp:21-77
    def __init__(self, alias_lookup):
        self.alias = alias_lookup
        self.indexed_values: Dict[str, int] = {}
        self.indexed_sublineage_values: Dict[str, int] = {}
        self.row_count = 0
        self.values: List[str] = []  # unaliased value per row (value_ids + lookup_unaliased)

    def lookup_aliased_value(self, row):  # lookupAliasedValue(getValues()[row]) :86-88
        return self.alias.alias(self.values[row])

    def insert(self, value):
        resolved = self.alias.unalias(value if value is not None else "")
        row = self.row_count
        self.row_count += 1
        self.values.append(resolved)
        self.indexed_values[resolved] = self.indexed_values.get(resolved, 0) | (1 << row)
        for parent in get_parent_lineages(resolved):
            self.indexed_sublineage_values[parent] = self.indexed_sublineage_values.get(parent, 0) | (1 << row)

    def filter(self, value):
        return self.indexed_values.get(self.alias.unalias(value))

    def filter_including_sublineages(self, value):
        return self.indexed_sublineage_values.get(self.alias.unalias(value))


@dataclass
class DatabasePartition:  # database_partition.h:39-112
    sequence_count: int = 0
    nuc_sequences: Dict[str, SequenceStorePartition] = field(default_factory=dict)
    aa_sequences: Dict[str, SequenceStorePartition] = field(default_factory=dict)
    pango_lineage_columns: Dict[str, PangoLineageColumnPartition] = field(default_factory=dict)
    # metadata columns (column_group.h): name -> list of raw values; nulls are "" / INT32_MIN / NaN / date 0
    string_columns: Dict[str, List[str]] = field(default_factory=dict)
    indexed_string_columns: Dict[str, List[str]] = field(default_factory=dict)
    int_columns: Dict[str, List[int]] = field(default_factory=dict)
    float_columns: Dict[str, List[float]] = field(default_factory=dict)
    date_columns: Dict[str, List[int]] = field(default_factory=dict)
    unaligned_nuc_sequences: Dict[str, List[Optional[str]]] = field(default_factory=dict)  # unaligned_sequence_store.h
    insertion_columns: Dict[str, List[str]] = field(default_factory=dict)  # standardised text per row (lookupValue)
    # insertion_column.cpp / insertion_index.cpp: column -> sequence name -> position -> insertion -> set of rows
    nuc_insertion_indexes: Dict[str, Dict[str, Dict[int, Dict[str, set]]]] = field(default_factory=dict)
    aa_insertion_indexes: Dict[str, Dict[str, Dict[int, Dict[str, set]]]] = field(default_factory=dict)


INT32_MIN = -(1 << 31)


def string_to_date(value):  # common/date.cpp:22-69: year<<16 | month<<12 | day, 0 = NULL
    if not value:
        return 0
    first = value.find("-")
    if first == -1:
        return 0
    second = value.find("-", first + 1)
    if second == -1:
        return 0

    def stoi(text):  # std::stoi: optional whitespace / sign, then leading digits
        import re

        match = re.match(r"\s*[+-]?\d+", text)
        if match is None:
            raise ValueError(text)
        return int(match.group(0))

    try:
        year, month, day = stoi(value[:first]), stoi(value[first + 1 :]), stoi(value[second + 1 :])
    except ValueError:
        return 0
    if month > 12 or month == 0 or day > 31 or day == 0:
        return 0
    return ((year << 16) + (month << 12) + day) & UINT32


def date_to_string(date):  # common/date.cpp:71-86
    if date == 0:
        return None
    return "%04d-%02d-%02d" % (date >> 16, (date >> 12) & 0xF, date & 0xFFF)


class Database:
    """The slice of silo::Database the path reads (database.h:55-79)."""

    def __init__(self, nuc_references, aa_references, default_nucleotide_sequence="main", alias_key=None):
        self.nuc_references = {k: list(v) for k, v in nuc_references.items()}  # name -> symbol ids
        self.aa_references = {k: list(v) for k, v in aa_references.items()}
        self.default_nucleotide_sequence = default_nucleotide_sequence
        self.alias = PangoLineageAliasLookup(alias_key or {})
        self.partitions: List[DatabasePartition] = []
        # database_config.yaml: [(name, column type)] in file order, primaryKey, dateToSortBy
        self.metadata: List = []
        self.primary_key = None
        self.date_to_sort_by = None

    COLUMN_TYPES = ("string", "indexed_string", "pango_lineage", "date", "int", "float", "insertion", "aa_insertion")

    def set_config(self, metadata, primary_key, date_to_sort_by=None):
        """metadata: [(name, type)] with type in COLUMN_TYPES (database_config.cpp:158-189 getColumnType)."""
        self.metadata = list(metadata)
        self.primary_key = primary_key
        self.date_to_sort_by = date_to_sort_by

    def column_type(self, name):
        for column, kind in self.metadata:
            if column == name:
                return kind
        return None

    def add_metadata(self, partition, rows):
        """rows: list of dict column -> raw text ('' = null), as the metadata TSV holds them
        (column inserts: string_column.cpp:16-24, indexed_string_column.cpp:24-36, int_column.cpp:17-24,
        float_column.cpp:16-24, date_column.cpp:15-21)."""
        assert len(rows) == partition.sequence_count
        for name, kind in self.metadata:
            raw = [row.get(name) or "" for row in rows]
            if kind == "string":
                partition.string_columns[name] = raw
            elif kind == "indexed_string":
                partition.indexed_string_columns[name] = raw
            elif kind == "pango_lineage":
                if name not in partition.pango_lineage_columns:
                    column = PangoLineageColumnPartition(self.alias)
                    for value in raw:
                        column.insert(value)
                    partition.pango_lineage_columns[name] = column
            elif kind == "int":
                partition.int_columns[name] = [INT32_MIN if value == "" else int(value) for value in raw]
            elif kind == "float":
                partition.float_columns[name] = [float("nan") if value == "" else float(value) for value in raw]
            elif kind == "date":
                partition.date_columns[name] = [string_to_date(value) for value in raw]
            elif kind in ("insertion", "aa_insertion"):
                # InsertionColumnPartition::insert (insertion_column.cpp:76-113): "position:insertion" entries belong to
                # the default sequence (the default nucleotide sequence; amino-acid columns have none), others name theirs
                default_sequence = self.default_nucleotide_sequence if kind == "insertion" else None
                index = {}
                standardised = []
                for row, value in enumerate(raw):
                    parts_out = []
                    for entry in (value.split(",") if value else []):
                        parts = [part.replace('"', "") for part in entry.split(":")]
                        if len(parts) == 2 and default_sequence is not None:
                            sequence_name, position, insertion = default_sequence, int(parts[0]), parts[1]
                        elif len(parts) == 3:
                            sequence_name, position, insertion = parts[0], int(parts[1]), parts[2]
                        else:
                            raise ValueError("Failed to parse insertion due to invalid format: " + entry)
                        index.setdefault(sequence_name, {}).setdefault(position, {}).setdefault(insertion, set()).add(row)
                        if default_sequence is not None and default_sequence == sequence_name:
                            parts_out.append(f"{position}:{insertion}")
                        else:
                            parts_out.append(f"{sequence_name}:{position}:{insertion}")
                    standardised.append(",".join(parts_out))
                partition.insertion_columns[name] = standardised
                (partition.nuc_insertion_indexes if kind == "insertion" else partition.aa_insertion_indexes)[name] = index
            else:
                raise ValueError(kind)

    def add_partition(self, nuc_genomes, aa_genomes, lineages=None, lineage_column="pango_lineage"):
        """nuc_genomes / aa_genomes: name -> list of str|None, all the same length."""
        partition = DatabasePartition()
        counts = set()
        for name, genomes in nuc_genomes.items():
            store = SequenceStorePartition(Nucleotide, self.nuc_references[name])
            store.fill(genomes)
            partition.nuc_sequences[name] = store
            counts.add(store.sequence_count)
        for name, genomes in aa_genomes.items():
            store = SequenceStorePartition(AminoAcid, self.aa_references[name])
            store.fill(genomes)
            partition.aa_sequences[name] = store
            counts.add(store.sequence_count)
        if lineages is not None:
            column = PangoLineageColumnPartition(self.alias)
            for value in lineages:
                column.insert(value)
            partition.pango_lineage_columns[lineage_column] = column
            counts.add(len(lineages))
        assert len(counts) == 1, counts
        partition.sequence_count = counts.pop()
        self.partitions.append(partition)
        return partition

    def sequence_store_names(self, alphabet):
        return self.nuc_references if alphabet is Nucleotide else self.aa_references


# ------------------------------------------------------------------------------------------------
# operators (operators/*.cpp); evaluate() returns an int bitset
# ------------------------------------------------------------------------------------------------
EMPTY, FULL, INDEX_SCAN, INTERSECTION, COMPLEMENT, UNION, THRESHOLD, BITMAP_SELECTION, SELECTION, RANGE_SELECTION = range(10)


class Operator:
    type = None

    def evaluate(self):
        raise NotImplementedError

    def copy(self):
        raise NotImplementedError

    def negate(self):
        raise NotImplementedError


class Empty(Operator):
    type = EMPTY

    def __init__(self, row_count):
        self.row_count = row_count

    def evaluate(self):
        return 0

    def copy(self):
        return Empty(self.row_count)

    def negate(self):
        return Full(self.row_count)


class Full(Operator):
    type = FULL

    def __init__(self, row_count):
        self.row_count = row_count

    def evaluate(self):  # full.cpp:24-28 addRange(0,row_count)
        return (1 << self.row_count) - 1

    def copy(self):
        return Full(self.row_count)

    def negate(self):
        return Empty(self.row_count)


class IndexScan(Operator):
    type = INDEX_SCAN

    def __init__(self, bitmap, row_count):
        self.bitmap = bitmap
        self.row_count = row_count

    def evaluate(self):  # index_scan.cpp:28-30
        return self.bitmap

    def copy(self):
        return IndexScan(self.bitmap, self.row_count)

    def negate(self):
        return Complement(self.copy(), self.row_count)


class Complement(Operator):
    type = COMPLEMENT

    def __init__(self, child, row_count):
        self.child = child
        self.row_count = row_count

    @staticmethod
    def from_de_morgan(disjunction, row_count):  # complement.cpp:22-40
        non_negated, negated = [], []
        for child in disjunction:
            if child.type == COMPLEMENT:
                negated.append(child.negate())
            else:
                non_negated.append(child)
        return Complement(Intersection(negated, non_negated, row_count), row_count)

    def evaluate(self):  # complement.cpp:50-54
        return flip(self.child.evaluate(), 0, self.row_count)

    def copy(self):
        return Complement(self.child.copy(), self.row_count)

    def negate(self):
        return self.child.copy()


class Intersection(Operator):
    type = INTERSECTION

    def __init__(self, children, negated_children, row_count):  # intersection.cpp:22-47
        self.children = list(children)
        self.negated_children = list(negated_children)
        self.row_count = row_count
        if not self.children:
            raise QueryCompilationException(
                "Compilation bug: Intersection without non-negated children is not allowed. "
                "Should be compiled as a union."
            )
        if len(self.children) + len(self.negated_children) < 2:
            raise QueryCompilationException("Compilation bug: Intersection needs at least two children.")

    def evaluate(self):  # intersection.cpp:80-127
        children = sorted((c.evaluate() for c in self.children), key=card)
        negated = sorted((c.evaluate() for c in self.negated_children), key=card, reverse=True)
        result = children[0]
        for bitmap in children[1:]:
            result &= bitmap
        for bitmap in negated:
            result &= ~bitmap
        return result

    def copy(self):
        return Intersection([c.copy() for c in self.children], [c.copy() for c in self.negated_children], self.row_count)

    def negate(self):
        return Complement(self.copy(), self.row_count)


class Union(Operator):
    type = UNION

    def __init__(self, children, row_count):
        self.children = list(children)
        self.row_count = row_count

    def evaluate(self):  # union.cpp:35-45
        result = 0
        for child in self.children:
            result |= child.evaluate()
        return result

    def copy(self):
        return Union([c.copy() for c in self.children], self.row_count)

    def negate(self):
        return Complement(self.copy(), self.row_count)


class Threshold(Operator):
    type = THRESHOLD

    def __init__(self, non_negated_children, negated_children, number_of_matchers, match_exactly, row_count):
        self.non_negated_children = list(non_negated_children)
        self.negated_children = list(negated_children)
        self.number_of_matchers = number_of_matchers
        self.match_exactly = match_exactly
        self.row_count = row_count
        if number_of_matchers >= len(self.non_negated_children) + len(self.negated_children):  # threshold.cpp:28-33
            raise QueryCompilationException(
                "Compilation Error: number_of_matchers must be less than the number of children of a "
                "threshold expression"
            )
        if number_of_matchers == 0:
            raise QueryCompilationException("Compilation Error: number_of_matchers must be greater than zero")

    def evaluate(self):  # threshold.cpp:64-138, the DP table verbatim
        n = self.number_of_matchers
        table_size = n + 1 if self.match_exactly else n
        table = [0] * table_size
        if not self.non_negated_children:
            table[0] = flip(self.negated_children[0].evaluate(), 0, self.row_count)
        else:
            table[0] = self.non_negated_children[0].evaluate()
        max_index = table_size - 1
        non_negated_count = len(self.non_negated_children)
        negated_count = len(self.negated_children)
        k = non_negated_count + negated_count
        for i in range(1, non_negated_count):
            bitmap = self.non_negated_children[i].evaluate()
            for j in range(min(max_index, i), max(0, n - k + i - 1), -1):
                table[j] |= table[j - 1] & bitmap
            if k - i > n - 1:
                table[0] |= bitmap
        took_first = 1 if not self.non_negated_children else 0
        for local_i in range(took_first, negated_count):
            bitmap = self.negated_children[local_i].evaluate()
            i = local_i + non_negated_count
            for j in range(min(max_index, i), max(0, n - k + i - 1), -1):
                table[j] |= table[j - 1] & ~bitmap
            if k - i > n - 1:
                table[0] |= flip(bitmap, 0, self.row_count)
        if self.match_exactly:
            return table[n - 1] & ~table[n]
        return table[-1]

    def copy(self):
        return Threshold(
            [c.copy() for c in self.non_negated_children], [c.copy() for c in self.negated_children],
            self.number_of_matchers, self.match_exactly, self.row_count,
        )

    def negate(self):
        return Complement(self.copy(), self.row_count)


class BitmapSelection(Operator):
    type = BITMAP_SELECTION
    CONTAINS, NOT_CONTAINS = 0, 1

    def __init__(self, bitmaps, row_count, comparator, value):
        self.bitmaps = bitmaps  # row-wise list of sets
        self.row_count = row_count
        self.comparator = comparator
        self.value = value

    def evaluate(self):  # bitmap_selection.cpp:33-52
        out = 0
        for i in range(self.row_count):
            contains = self.value in self.bitmaps[i]
            if contains == (self.comparator == self.CONTAINS):
                out |= 1 << i
        return out

    def copy(self):
        return BitmapSelection(self.bitmaps, self.row_count, self.comparator, self.value)

    def negate(self):
        flipped = self.NOT_CONTAINS if self.comparator == self.CONTAINS else self.CONTAINS
        return BitmapSelection(self.bitmaps, self.row_count, flipped, self.value)


# ------------------------------------------------------------------------------------------------
# filter expressions (filter_expressions/*.cpp)
# ------------------------------------------------------------------------------------------------
NONE, UPPER_BOUND, LOWER_BOUND = "NONE", "UPPER_BOUND", "LOWER_BOUND"


def invert_mode(mode):  # expression.cpp:38-46
    if mode == UPPER_BOUND:
        return LOWER_BOUND
    if mode == LOWER_BOUND:
        return UPPER_BOUND
    return mode


class Expression:
    def compile(self, database, partition, mode):
        raise NotImplementedError


class TrueExpr(Expression):
    def compile(self, database, partition, mode):
        return Full(partition.sequence_count)


class FalseExpr(Expression):
    def compile(self, database, partition, mode):
        return Empty(partition.sequence_count)


class And(Expression):
    def __init__(self, children):
        self.children = children

    def compile_children(self, database, partition, mode):  # and.cpp:101-172
        all_children = [c.compile(database, partition, mode) for c in self.children]
        non_negated, negated, predicates = [], [], []
        index = 0
        while index < len(all_children):  # the list grows while it is walked (:144-155, intended behaviour)
            child = all_children[index]
            index += 1
            if child.type == FULL:
                continue
            if child.type == EMPTY:
                return [Empty(partition.sequence_count)], [], []
            if child.type == INTERSECTION:
                non_negated.extend(child.children)
                negated.extend(child.negated_children)
            elif child.type == COMPLEMENT:
                negated.append(child.negate())
            elif child.type == SELECTION:
                predicates.extend(child.predicates)
                if child.child is not None:
                    all_children.append(child.child)
            else:
                non_negated.append(child)
        return non_negated, negated, predicates

    def compile(self, database, partition, mode):  # and.cpp:174-227
        non_negated, negated, predicates = self.compile_children(database, partition, mode)
        row_count = partition.sequence_count
        if not non_negated and not negated:
            if not predicates:
                return Full(row_count)
            return Selection(predicates, row_count)
        if len(non_negated) == 1 and not negated:
            index_operator = non_negated[0]
        elif len(negated) == 1 and not non_negated:
            index_operator = Complement(negated[0], row_count)
        elif not non_negated:
            index_operator = Complement(Union(negated, row_count), row_count)
        else:
            index_operator = Intersection(non_negated, negated, row_count)
        if not predicates:
            return index_operator
        return Selection(predicates, row_count, index_operator)


class Or(Expression):
    def __init__(self, children):
        self.children = children

    def compile(self, database, partition, mode):  # or.cpp:41-94
        row_count = partition.sequence_count
        all_children = [c.compile(database, partition, mode) for c in self.children]
        filtered = []
        for child in all_children:
            if child.type == EMPTY:
                continue
            if child.type == FULL:
                return Full(row_count)
            if child.type == UNION:
                filtered.extend(child.children)
            else:
                filtered.append(child)
        if not filtered:
            return Empty(row_count)
        if len(filtered) == 1:
            return filtered[0]
        if any(child.type == COMPLEMENT for child in filtered):
            return Complement.from_de_morgan(filtered, row_count)
        return Union(filtered, row_count)


class Negation(Expression):
    def __init__(self, child):
        self.child = child

    def compile(self, database, partition, mode):  # negation.cpp:27-34
        return self.child.compile(database, partition, invert_mode(mode)).negate()


class Maybe(Expression):
    def __init__(self, child):
        self.child = child

    def compile(self, database, partition, mode):  # maybe.cpp:26-32
        return self.child.compile(database, partition, UPPER_BOUND)


class Exact(Expression):
    def __init__(self, child):
        self.child = child

    def compile(self, database, partition, mode):  # exact.cpp:26-32
        return self.child.compile(database, partition, LOWER_BOUND)


def _nof_trivial(n, non_negated, negated, match_exactly, row_count):  # nof.cpp:35-88
    count = len(non_negated) + len(negated)
    if n > count:
        return Empty(row_count)
    if n < 0:
        return Empty(row_count) if match_exactly else Full(row_count)
    if n == 0:
        if not match_exactly:
            return Full(row_count)
        if count == 0:
            return Full(row_count)
        if count == 1:
            if not non_negated:
                return negated[0]
            return Complement(non_negated[0], row_count)
        if not negated:
            return Complement(Union(non_negated, row_count), row_count)
        return Intersection(negated, non_negated, row_count)
    if n == 1 and count == 1:
        if not negated:
            return non_negated[0]
        return Complement(negated[0], row_count)
    return None


def _nof_to_operator(n, non_negated, negated, match_exactly, row_count):  # nof.cpp:120-154
    trivial = _nof_trivial(n, non_negated, negated, match_exactly, row_count)
    if trivial is not None:
        return trivial
    count = len(non_negated) + len(negated)
    if n == count:  # handleAndCase :90-103
        if not non_negated:
            return Complement(Union(negated, row_count), row_count)
        return Intersection(non_negated, negated, row_count)
    if n == 1 and not match_exactly:  # handleOrCase :105-118
        if not negated:
            return Union(non_negated, row_count)
        return Complement(Intersection(negated, non_negated, row_count), row_count)
    return Threshold(non_negated, negated, n, match_exactly, row_count)


class NOf(Expression):
    def __init__(self, children, number_of_matchers, match_exactly):
        self.children = children
        self.number_of_matchers = number_of_matchers
        self.match_exactly = match_exactly

    def map_child_expressions(self, database, partition, mode):  # nof.cpp:185-218
        non_negated, negated = [], []
        n = self.number_of_matchers
        for child in self.children:
            op = child.compile(database, partition, mode)
            if op.type == EMPTY:
                continue
            if op.type == FULL:
                n -= 1
            elif op.type == COMPLEMENT:
                negated.append(op.negate())
            else:
                non_negated.append(op)
        return non_negated, negated, n

    def rewrite_non_exact(self, database, partition, mode):  # nof.cpp:220-258
        row_count = partition.sequence_count
        non_negated, negated, n = self.map_child_expressions(database, partition, mode)
        at_least_k = [_nof_to_operator(n, non_negated, negated, False, row_count)]
        non_negated, negated, n = self.map_child_expressions(database, partition, mode)
        at_least_k_plus_one = [_nof_to_operator(n + 1, non_negated, negated, False, row_count)]
        return _nof_to_operator(2, at_least_k, at_least_k_plus_one, False, row_count)

    def compile(self, database, partition, mode):  # nof.cpp:260-280
        non_negated, negated, n = self.map_child_expressions(database, partition, mode)
        if mode != NONE and self.match_exactly and self.number_of_matchers < len(self.children):
            return self.rewrite_non_exact(database, partition, mode)
        return _nof_to_operator(n, non_negated, negated, self.match_exactly, partition.sequence_count)


class NucleotideSymbolEquals(Expression):
    def __init__(self, sequence_name, position, value):
        self.sequence_name = sequence_name
        self.position = position  # 0-based
        self.value = value  # symbol id or None ('.')

    def compile(self, database, partition, mode):  # nucleotide_symbol_equals.cpp:94-189
        name = self.sequence_name if self.sequence_name is not None else database.default_nucleotide_sequence
        check_silo_query(
            name in database.nuc_references,
            "Database does not contain the nucleotide sequence with name: '" + name + "'",
        )
        store = partition.nuc_sequences[name]
        if self.position >= len(store.reference_sequence):
            raise QueryParseException(
                "NucleotideEquals position is out of bounds '" + str(self.position + 1) + "' > '"
                + str(len(store.reference_sequence)) + "'"
            )
        symbol = self.value if self.value is not None else store.reference_sequence[self.position]
        row_count = partition.sequence_count
        if mode == UPPER_BOUND:
            filters = [NucleotideSymbolEquals(name, self.position, s) for s in AMBIGUITY_NUC_SYMBOLS[symbol]]
            return Or(filters).compile(database, partition, NONE)
        if symbol == Nucleotide.SYMBOL_MISSING:
            return BitmapSelection(
                store.missing_symbol_bitmaps, len(store.missing_symbol_bitmaps), BitmapSelection.CONTAINS, self.position
            )
        position = store.positions[self.position]
        if position.flipped == symbol:
            return Complement(IndexScan(position.bitmaps[symbol], row_count), row_count)
        if position.deleted == symbol:
            symbols = std_remove(Nucleotide.SYMBOLS, symbol)
            filters = [Negation(NucleotideSymbolEquals(name, self.position, s)) for s in symbols]
            return And(filters).compile(database, partition, NONE)
        return IndexScan(position.bitmaps[symbol], row_count)


class AASymbolEquals(Expression):
    def __init__(self, sequence_name, position, value):
        self.sequence_name = sequence_name
        self.position = position
        self.value = value

    def compile(self, database, partition, mode):  # aa_symbol_equals.cpp:41-92 (mode ignored)
        store = partition.aa_sequences[self.sequence_name]  # .at(): KeyError -> 500 in the reference
        if self.position >= len(store.reference_sequence):
            raise QueryParseException(
                "AminoAcidEquals position is out of bounds '" + str(self.position + 1) + "' > '"
                + str(len(store.reference_sequence)) + "'"
            )
        symbol = self.value if self.value is not None else store.reference_sequence[self.position]
        row_count = partition.sequence_count
        if symbol == AminoAcid.SYMBOL_MISSING:
            return BitmapSelection(
                store.missing_symbol_bitmaps, len(store.missing_symbol_bitmaps), BitmapSelection.CONTAINS, self.position
            )
        position = store.positions[self.position]
        if position.flipped == symbol:
            return Complement(IndexScan(position.bitmaps[symbol], row_count), row_count)
        if position.deleted == symbol:
            symbols = std_remove(AminoAcid.SYMBOLS, symbol)
            if symbol == AminoAcid.STOP:
                # Reference bug: STOP is last in SYMBOLS, so std::remove is a no-op
                # and the rewrite contains itself -> unbounded recursion.  Excluded from parity; the
                # intended set (complement of all other symbols) is returned instead.
                symbols = [s for s in AminoAcid.SYMBOLS if s != symbol]
            filters = [Negation(AASymbolEquals(self.sequence_name, self.position, s)) for s in symbols]
            return And(filters).compile(database, partition, NONE)
        return IndexScan(position.bitmaps[symbol], row_count)


class HasMutation(Expression):
    def __init__(self, sequence_name, position):
        self.sequence_name = sequence_name
        self.position = position

    def compile(self, database, partition, mode):  # has_mutation.cpp:35-78
        name = self.sequence_name if self.sequence_name is not None else database.default_nucleotide_sequence
        check_silo_query(
            name in database.nuc_references,
            "Database does not contain the nucleotide sequence with name: '" + name + "'",
        )
        ref_symbol = database.nuc_references[name][self.position]  # .at(): out of range -> 500
        if mode == UPPER_BOUND:
            return Negation(NucleotideSymbolEquals(name, self.position, ref_symbol)).compile(database, partition, NONE)
        symbols = std_remove([1, 2, 3, 4], ref_symbol)  # quirk: ref T keeps T (remove without erase)
        return Or([NucleotideSymbolEquals(name, self.position, s) for s in symbols]).compile(database, partition, NONE)


class HasAAMutation(Expression):
    def __init__(self, sequence_name, position):
        self.sequence_name = sequence_name
        self.position = position

    def compile(self, database, partition, mode):  # has_aa_mutation.cpp:33-63
        ref_symbol = database.aa_references[self.sequence_name][self.position]
        if mode == UPPER_BOUND:
            return Negation(AASymbolEquals(self.sequence_name, self.position, ref_symbol)).compile(database, partition, NONE)
        symbols = std_remove(std_remove(AminoAcid.SYMBOLS, AminoAcid.X), ref_symbol)
        return Or([AASymbolEquals(self.sequence_name, self.position, s) for s in symbols]).compile(database, partition, NONE)


class PangoLineageFilter(Expression):
    def __init__(self, column, lineage, include_sublineages):
        self.column = column
        self.lineage = lineage
        self.include_sublineages = include_sublineages

    def compile(self, database, partition, mode):  # pango_lineage_filter.cpp:37-59
        row_count = partition.sequence_count
        if self.column not in partition.pango_lineage_columns:
            return Empty(row_count)
        upper = self.lineage.upper()
        column = partition.pango_lineage_columns[self.column]
        bitmap = column.filter_including_sublineages(upper) if self.include_sublineages else column.filter(upper)
        if bitmap is None:
            return Empty(row_count)
        return IndexScan(bitmap, row_count)


# ---- JSON -> expression (the from_json functions) ----------------------------------------------------
def _is_unsigned(value):
    return isinstance(value, int) and not isinstance(value, bool) and value >= 0


def parse_expression(node):  # expression.cpp:49-102
    check_silo_query(isinstance(node, dict) and "type" in node, "The field 'type' is required in any filter expression")
    check_silo_query(
        isinstance(node["type"], str),
        "The field 'type' in all filter expressions needs to be a string, but is: " + json.dumps(node["type"]),
    )
    kind = node["type"]
    if kind == "True":
        return TrueExpr()
    if kind == "False":
        return FalseExpr()
    if kind in ("And", "Or"):
        label = "an " + kind
        check_silo_query("children" in node, f"The field 'children' is required in {label} expression")
        check_silo_query(isinstance(node["children"], list), f"The field 'children' in {label} expression needs to be an array")
        children = [parse_expression(child) for child in node["children"]]
        return And(children) if kind == "And" else Or(children)
    if kind == "N-Of":  # nof.cpp:283-316
        check_silo_query("children" in node, "The field 'children' is required in an N-Of expression")
        check_silo_query(isinstance(node["children"], list), "The field 'children' in an N-Of expression needs to be an array")
        check_silo_query("numberOfMatchers" in node, "The field 'numberOfMatchers' is required in an N-Of expression")
        check_silo_query(
            _is_unsigned(node["numberOfMatchers"]),
            "The field 'numberOfMatchers' in an N-Of expression needs to be an unsigned integer",
        )
        check_silo_query("matchExactly" in node, "The field 'matchExactly' is required in an N-Of expression")
        check_silo_query(isinstance(node["matchExactly"], bool), "The field 'matchExactly' in an N-Of expression needs to be a boolean")
        children = [parse_expression(child) for child in node["children"]]
        return NOf(children, node["numberOfMatchers"], node["matchExactly"])
    if kind == "Not":
        check_silo_query("child" in node, "The field 'child' is required in a Not expression")
        return Negation(parse_expression(node["child"]))
    if kind == "Maybe":
        check_silo_query("child" in node, "The field 'child' is required in a Maybe expression")
        return Maybe(parse_expression(node["child"]))
    if kind == "Exact":
        check_silo_query("child" in node, "The field 'child' is required in a Exact expression")
        return Exact(parse_expression(node["child"]))
    if kind == "NucleotideEquals":  # nucleotide_symbol_equals.cpp:192-227
        check_silo_query("position" in node, "The field 'position' is required in a NucleotideEquals expression")
        check_silo_query(
            _is_unsigned(node["position"]) and node["position"] > 0,
            "The field 'position' in a NucleotideEquals expression needs to be an unsigned integer greater than 0",
        )
        check_silo_query("symbol" in node, "The field 'symbol' is required in a NucleotideEquals expression")
        check_silo_query(isinstance(node["symbol"], str), "The field 'symbol' in a NucleotideEquals expression needs to be a string")
        name = node.get("sequenceName")
        symbol = node["symbol"]
        check_silo_query(len(symbol) == 1, "The string field 'symbol' must be exactly one character long")
        value = Nucleotide.char_to_symbol(symbol)
        check_silo_query(
            value is not None or symbol == ".",
            "The string field 'symbol' must be either a valid nucleotide symbol or the '.' symbol.",
        )
        return NucleotideSymbolEquals(name, node["position"] - 1, value)
    if kind == "AminoAcidEquals":  # aa_symbol_equals.cpp:95-125
        check_silo_query(
            isinstance(node.get("sequenceName"), str), "AminoAcidEquals expression requires the string field sequenceName"
        )
        check_silo_query("position" in node, "The field 'position' is required in a AminoAcidEquals expression")
        check_silo_query(
            _is_unsigned(node["position"]) and node["position"] > 0,
            "The field 'position' in a AminoAcidEquals expression needs to be an unsigned integer greater than 0",
        )
        check_silo_query(
            isinstance(node.get("symbol"), str), "The string field 'symbol' is required in a AminoAcidEquals expression"
        )
        symbol = node["symbol"]
        check_silo_query(len(symbol) == 1, "The string field 'symbol' must be exactly one character long")
        value = AminoAcid.char_to_symbol(symbol)
        check_silo_query(
            value is not None or symbol == ".",
            "The string field 'symbol' must be either a valid amino acid or the '.' symbol.",
        )
        return AASymbolEquals(node["sequenceName"], node["position"] - 1, value)
    if kind == "HasNucleotideMutation":  # has_mutation.cpp:81-96 (position 0 is NOT rejected here)
        check_silo_query("position" in node, "The field 'position' is required in a HasNucleotideMutation expression")
        check_silo_query(
            _is_unsigned(node["position"]),
            "The field 'position' in a HasNucleotideMutation expression needs to be an unsigned integer",
        )
        return HasMutation(node.get("sequenceName"), (node["position"] - 1) & UINT32)
    if kind == "HasAminoAcidMutation":  # has_aa_mutation.cpp:66-84
        check_silo_query("position" in node, "The field 'position' is required in a HasAminoAcidMutation expression")
        check_silo_query(
            _is_unsigned(node["position"]),
            "The field 'position' in a HasAminoAcidMutation expression needs to be an unsigned integer",
        )
        check_silo_query(
            isinstance(node.get("sequenceName"), str),
            "HasAminoAcidMutation expression requires the string field sequenceName",
        )
        return HasAAMutation(node["sequenceName"], (node["position"] - 1) & UINT32)
    if kind == "PangoLineage":  # pango_lineage_filter.cpp:62-92
        check_silo_query("column" in node, "The field 'column' is required in a PangoLineage expression")
        check_silo_query(isinstance(node["column"], str), "The field 'column' in a PangoLineage expression needs to be a string")
        check_silo_query("value" in node, "The field 'value' is required in a PangoLineage expression")
        check_silo_query(isinstance(node["value"], str), "The field 'value' in a PangoLineage expression needs to be a string")
        check_silo_query("includeSublineages" in node, "The field 'includeSublineages' is required in a PangoLineage expression")
        check_silo_query(
            isinstance(node["includeSublineages"], bool),
            "The field 'includeSublineages' in a PangoLineage expression needs to be a boolean",
        )
        return PangoLineageFilter(node["column"], node["value"], node["includeSublineages"])
    if kind == "StringEquals":  # string_equals.cpp:70-85
        check_silo_query("column" in node, "The field 'column' is required in an StringEquals expression")
        check_silo_query(isinstance(node["column"], str), "The field 'column' in an StringEquals expression needs to be a string")
        check_silo_query("value" in node, "The field 'value' is required in an StringEquals expression")
        check_silo_query(
            node["value"] is None or isinstance(node["value"], str),
            "The field 'value' in an StringEquals expression needs to be a string or null",
        )
        return StringEquals(node["column"], node["value"] if node["value"] is not None else "")
    if kind == "IntEquals":  # int_equals.cpp:50-67
        check_silo_query("column" in node, "The field 'column' is required in an IntEquals expression")
        check_silo_query(isinstance(node["column"], str), "The field 'column' in an IntEquals expression must be a string")
        check_silo_query("value" in node, "The field 'value' is required in an IntEquals expression")
        check_silo_query(node["value"] is None or _is_integer(node["value"]), "The field 'value' in an IntEquals expression must be an integer or null")
        return IntEquals(node["column"], INT32_MIN if node["value"] is None else node["value"])
    if kind == "IntBetween":  # int_between.cpp:63-88
        check_silo_query("column" in node, "The field 'column' is required in a IntBetween expression")
        check_silo_query(isinstance(node["column"], str), "The field 'column' in a IntBetween expression must be a string")
        check_silo_query("from" in node, "The field 'from' is required in IntBetween expression")
        check_silo_query(node["from"] is None or _is_integer(node["from"]), "The field 'from' in a IntBetween expression must be an int or null")
        check_silo_query("to" in node, "The field 'to' is required in a IntBetween expression")
        check_silo_query(node["to"] is None or _is_integer(node["to"]), "The field 'to' in a IntBetween expression must be an int or null")
        return IntBetween(node["column"], node["from"], node["to"])
    if kind == "FloatEquals":  # float_equals.cpp:53-70
        check_silo_query("column" in node, "The field 'column' is required in an FloatEquals expression")
        check_silo_query(isinstance(node["column"], str), "The field 'column' in an FloatEquals expression must be a string")
        check_silo_query("value" in node, "The field 'value' is required in an FloatEquals expression")
        check_silo_query(node["value"] is None or isinstance(node["value"], float), "The field 'value' in an FloatEquals expression must be a float")
        return FloatEquals(node["column"], float("nan") if node["value"] is None else node["value"])
    if kind == "FloatBetween":  # float_between.cpp:72-99
        check_silo_query("column" in node, "The field 'column' is required in a FloatBetween expression")
        check_silo_query(isinstance(node["column"], str), "The field 'column' in a FloatBetween expression must be a string")
        check_silo_query("from" in node, "The field 'from' is required in FloatBetween expression")
        check_silo_query(node["from"] is None or isinstance(node["from"], float), "The field 'from' in a FloatBetween expression must be a float or null")
        check_silo_query("to" in node, "The field 'to' is required in a FloatBetween expression")
        check_silo_query(node["to"] is None or isinstance(node["to"], float), "The field 'to' in a FloatBetween expression must be a float or null")
        return FloatBetween(node["column"], node["from"], node["to"])
    if kind == "DateBetween":  # date_between.cpp:103-130
        check_silo_query("column" in node, "The field 'column' is required in a DateBetween expression")
        check_silo_query(isinstance(node["column"], str), "The field 'column' in a DateBetween expression needs to be a string")
        check_silo_query("from" in node, "The field 'from' is required in DateBetween expression")
        check_silo_query(
            node["from"] is None or (isinstance(node["from"], str) and True),
            "The field 'from' in a DateBetween expression needs to be a string or null",
        )
        check_silo_query("to" in node, "The field 'to' is required in a DateBetween expression")
        check_silo_query(
            node["to"] is None or (isinstance(node["to"], str) and True),
            "The field 'to' in a DateBetween expression needs to be a non-empty string or null",
        )
        return DateBetween(
            node["column"],
            string_to_date(node["from"]) if isinstance(node["from"], str) else None,
            string_to_date(node["to"]) if isinstance(node["to"], str) else None,
        )
    if kind == "InsertionContains":
        return parse_insertion_contains(node, Nucleotide)
    if kind == "AminoAcidInsertionContains":
        return parse_insertion_contains(node, AminoAcid)
    raise QueryParseException("Unknown object filter type '" + kind + "'")


def _is_integer(value):
    return isinstance(value, int) and not isinstance(value, bool)


# ------------------------------------------------------------------------------------------------
# metadata predicates (operators/selection.cpp, range_selection.cpp; filter_expressions/{string_equals,
# int_equals,int_between,float_equals,float_between,date_between}.cpp)
# ------------------------------------------------------------------------------------------------
EQUALS, NOT_EQUALS, LESS, HIGHER_OR_EQUALS, HIGHER, LESS_OR_EQUALS = "==", "!=", "<", ">=", ">", "<="
_NEGATED = {EQUALS: NOT_EQUALS, NOT_EQUALS: EQUALS, LESS: HIGHER_OR_EQUALS, HIGHER_OR_EQUALS: LESS, HIGHER: LESS_OR_EQUALS,
            LESS_OR_EQUALS: HIGHER}


class Predicate:  # CompareToValueSelection<T>, selection.cpp:132-230; Python floats compare like IEEE doubles (NaN: all false but !=)
    def __init__(self, column, comparator, value):
        self.column, self.comparator, self.value = column, comparator, value

    def match(self, row):
        v = self.column[row]
        c = self.comparator
        if c == EQUALS:
            return v == self.value
        if c == NOT_EQUALS:
            return v != self.value
        if c == LESS:
            return v < self.value
        if c == HIGHER_OR_EQUALS:
            return v >= self.value
        if c == HIGHER:
            return v > self.value
        return v <= self.value

    def negate(self):  # selection.cpp:195-220: the comparator is negated, NOT the truth value (differs for NaN)
        return Predicate(self.column, _NEGATED[self.comparator], self.value)


class Selection(Operator):
    type = SELECTION

    def __init__(self, predicates, row_count, child=None):
        self.predicates, self.row_count, self.child = list(predicates), row_count, child

    def evaluate(self):  # selection.cpp:88-108
        rows = range(self.row_count) if self.child is None else ids_from_bits(self.child.evaluate())
        return bits_from_ids([row for row in rows if all(p.match(row) for p in self.predicates)])

    def copy(self):
        return Selection(self.predicates, self.row_count, None if self.child is None else self.child.copy())

    def negate(self):  # selection.cpp:125-130
        if self.child is None and len(self.predicates) == 1:
            return Selection([self.predicates[0].negate()], self.row_count)
        return Complement(self.copy(), self.row_count)


class RangeSelection(Operator):  # range_selection.cpp: union of [start, end) id ranges
    type = RANGE_SELECTION

    def __init__(self, ranges, row_count):
        self.ranges, self.row_count = list(ranges), row_count

    def evaluate(self):
        bits = 0
        for start, end in self.ranges:
            if end > start:
                bits |= ((1 << (end - start)) - 1) << start
        return bits

    def copy(self):
        return RangeSelection(self.ranges, self.row_count)

    def negate(self):  # range_selection.cpp:44-66: the complementary ranges
        out, last = [], 0
        for start, end in self.ranges:
            if last != start:
                out.append((last, start))
            last = end
        if last != self.row_count:
            out.append((last, self.row_count))
        return RangeSelection(out, self.row_count)


class StringEquals(Expression):  # string_equals.cpp:37-68
    def __init__(self, column, value):
        self.column, self.value = column, value

    def compile(self, database, partition, mode):
        rows = partition.sequence_count
        if self.column in partition.indexed_string_columns:
            values = partition.indexed_string_columns[self.column]
            bitmap = bits_from_ids([i for i, v in enumerate(values) if v == self.value])
            return Empty(rows) if bitmap == 0 else IndexScan(bitmap, rows)
        if self.column in partition.string_columns:
            values = partition.string_columns[self.column]
            # embedString fails (-> Empty) only for a long string that is not in the dictionary: no row can equal it
            return Selection([Predicate(values, EQUALS, self.value)], rows)
        return Empty(rows)


class IntEquals(Expression):  # int_equals.cpp:30-47
    def __init__(self, column, value):
        self.column, self.value = column, value

    def compile(self, database, partition, mode):
        if self.column not in partition.int_columns:
            return Empty(partition.sequence_count)
        return Selection([Predicate(partition.int_columns[self.column], EQUALS, self.value)], partition.sequence_count)


class IntBetween(Expression):  # int_between.cpp:37-60 (.at(column): unknown column is a std::out_of_range -> 500)
    def __init__(self, column, value_from, value_to):
        self.column, self.value_from, self.value_to = column, value_from, value_to

    def compile(self, database, partition, mode):
        values = partition.int_columns[self.column]
        predicates = [Predicate(values, HIGHER_OR_EQUALS, self.value_from if self.value_from is not None else INT32_MIN + 1)]
        if self.value_to is not None:
            predicates.append(Predicate(values, LESS_OR_EQUALS, self.value_to))
        return Selection(predicates, partition.sequence_count)


class FloatEquals(Expression):  # float_equals.cpp:33-50
    def __init__(self, column, value):
        self.column, self.value = column, value

    def compile(self, database, partition, mode):
        if self.column not in partition.float_columns:
            return Empty(partition.sequence_count)
        return Selection([Predicate(partition.float_columns[self.column], EQUALS, self.value)], partition.sequence_count)


class FloatBetween(Expression):  # float_between.cpp:37-69: [from, to) — the upper bound is exclusive
    def __init__(self, column, value_from, value_to):
        self.column, self.value_from, self.value_to = column, value_from, value_to

    def compile(self, database, partition, mode):
        check_silo_query(self.column in partition.float_columns, "The database does not contain the float column '" + self.column + "'")
        values = partition.float_columns[self.column]
        predicates = []
        if self.value_from is not None:
            predicates.append(Predicate(values, HIGHER_OR_EQUALS, self.value_from))
        if self.value_to is not None:
            predicates.append(Predicate(values, LESS, self.value_to))
        if not predicates:
            predicates.append(Predicate(values, NOT_EQUALS, float("nan")))  # true for every row, NULLs included
        return Selection(predicates, partition.sequence_count)


class DateBetween(Expression):  # date_between.cpp:49-101
    def __init__(self, column, date_from, date_to):
        self.column, self.date_from, self.date_to = column, date_from, date_to

    def compile(self, database, partition, mode):
        values = partition.date_columns[self.column]
        rows = partition.sequence_count
        if database.date_to_sort_by != self.column:  # unsorted column: from <= d < to, the upper bound EXCLUSIVE
            return Selection(
                [Predicate(values, HIGHER_OR_EQUALS, self.date_from if self.date_from is not None else 1),
                 Predicate(values, LESS, self.date_to if self.date_to is not None else UINT32)], rows)
        # sorted column: lower_bound / upper_bound per chunk, i.e. from <= d <= to, the upper bound INCLUSIVE, NULL (0)
        # excluded.  The reference's rows are physically sorted by this column; here rows keep the input order, so the
        # id ranges are restated as the id set (same set, see SURVEY.md §8c on row order).
        low = self.date_from if self.date_from is not None else 1
        ids = [i for i, d in enumerate(values) if d >= low and (self.date_to is None or d <= self.date_to)]
        return IndexScan(bits_from_ids(ids), rows) if ids else RangeSelection([], rows)


class InsertionContains(Expression):  # insertion_contains.cpp:65-131
    def __init__(self, alphabet, column_names, sequence_name, position, value):
        self.alphabet, self.column_names, self.sequence_name, self.position, self.value = alphabet, column_names, sequence_name, position, value

    def compile(self, database, partition, mode):
        import re

        indexes = partition.nuc_insertion_indexes if self.alphabet is Nucleotide else partition.aa_insertion_indexes
        for column_name in self.column_names:
            check_silo_query(column_name in indexes, "The insertion column '" + column_name + "' does not exist.")
        rows = partition.sequence_count
        if not indexes:
            return Empty(rows)
        if self.sequence_name is not None:
            sequence_name = self.sequence_name
        else:
            default = database.default_nucleotide_sequence if self.alphabet is Nucleotide else None
            check_silo_query(default is not None, "The database has no default " + self.alphabet.NAME_LOWER + " sequence name")
            sequence_name = default
        operators = []
        for column_name in sorted(indexes):
            if self.column_names and column_name not in self.column_names:
                continue
            if sequence_name not in indexes[column_name]:
                continue
            # InsertionIndex::search (insertion_index.cpp:271-281): the 3-mer index only pre-selects candidates, the
            # answer is regex_search of the pattern over the distinct insertions at the position
            bits = 0
            for insertion, row_set in indexes[column_name][sequence_name].get(self.position, {}).items():
                if re.search(self.value, insertion):
                    bits |= bits_from_ids(sorted(row_set))
            operators.append(IndexScan(bits, rows))  # BitmapProducer: a computed bitmap
        if not operators:
            return Empty(rows)
        if len(operators) == 1:
            return operators[0]
        return Union(operators, rows)


def parse_insertion_contains(node, alphabet):  # insertion_contains.cpp:155-214
    import re

    check_silo_query(
        "column" not in node or isinstance(node["column"], (str, list)),
        "The InsertionsContains filter can have the field column of type string or an array of strings, but no other type",
    )
    column_names = []
    if isinstance(node.get("column"), list):
        for child in node["column"]:
            check_silo_query(
                isinstance(child, str),
                "The field column of the InsertionsContains filter must have type string or an array, if present. Found:"
                + json.dumps(child, separators=(",", ":")),
            )
            column_names.append(child)
    elif isinstance(node.get("column"), str):
        column_names.append(node["column"])
    check_silo_query("position" in node, "The field 'position' is required in an InsertionContains expression")
    check_silo_query(
        _is_unsigned(node["position"]) and node["position"] > 0,
        "The field 'position' in an InsertionContains expression needs to be a positive number (> 0)",
    )
    check_silo_query(
        "sequenceName" not in node or isinstance(node["sequenceName"], str),
        "The optional field 'sequenceName' in an InsertionContains expression needs to be a string",
    )
    check_silo_query("value" in node, "The field 'value' is required in an InsertionContains expression")
    check_silo_query(isinstance(node["value"], str), "The field 'value' in an InsertionContains expression needs to be a string")
    value = node["value"]
    check_silo_query(value != "", "The field 'value' in an InsertionContains expression must not be an empty string")
    symbols = "".join(alphabet.symbol_to_char(symbol) for symbol in alphabet.SYMBOLS)
    valid = re.compile("^([" + re.escape(symbols).replace("\\-", "-") + "]|\\.\\*)*$")  # ^([symbols]|\.\*)*$
    check_silo_query(
        valid.search(value) is not None,
        "The field 'value' in the InsertionContains expression does not contain a valid regex pattern: \"" + value
        + "\". It must only consist of " + alphabet.NAME_LOWER + " symbols and the regex symbol '.*'.",
    )
    return InsertionContains(alphabet, column_names, node.get("sequenceName"), node["position"], value)


# ------------------------------------------------------------------------------------------------
# actions (actions/action.cpp, aggregated.cpp, mutations.cpp)
# ------------------------------------------------------------------------------------------------
@dataclass
class OrderByField:
    name: str
    ascending: bool


class Action:
    def __init__(self):
        self.order_by_fields: List[OrderByField] = []
        self.limit: Optional[int] = None
        self.offset: Optional[int] = None

    def validate_order_by_fields(self, database):
        pass

    def execute(self, database, filters):
        raise NotImplementedError

    def execute_and_order(self, database, filters):  # action.cpp:104-117
        self.validate_order_by_fields(database)
        result = self.execute(database, filters)
        if self.offset is not None and self.offset >= len(result):
            return []
        result = self.apply_sort(result)
        return self.apply_offset_and_limit(result)

    def apply_sort(self, result):  # action.cpp:37-66; std::(partial_)sort is unstable: ties are unordered
        if not self.order_by_fields:
            return result
        import functools

        def key_of(value):  # optional<variant<string,int32,double>> ordering: nullopt < value; by index then value
            if value is None:
                return (0, 0, 0)
            if isinstance(value, str):
                return (1, 0, value)
            if isinstance(value, bool) or isinstance(value, int):
                return (1, 1, value)
            return (1, 2, value)

        def compare(a, b):
            for f in self.order_by_fields:
                ka, kb = key_of(a[f.name]), key_of(b[f.name])
                if ka == kb:
                    continue
                less = ka < kb
                return -1 if (less == f.ascending) else 1
            return 0

        return sorted(result, key=functools.cmp_to_key(compare))

    def apply_offset_and_limit(self, result):  # action.cpp:68-91
        limit = self.limit if self.limit is not None else len(result)
        end = min(limit + (self.offset or 0), len(result))
        if self.offset is not None and self.offset >= end:
            return []
        return result[(self.offset or 0) : end]


class Aggregated(Action):
    def __init__(self, group_by_fields):
        super().__init__()
        self.group_by_fields = group_by_fields

    def validate_order_by_fields(self, database):  # aggregated.cpp:26-38,71-88
        for name in self.group_by_fields:
            check_silo_query(database.column_type(name) is not None, "Metadata field '" + name + "' to group by not found")
        for f in self.order_by_fields:
            check_silo_query(
                f.name == "count" or f.name in self.group_by_fields,
                "The orderByField '" + f.name + "' cannot be ordered by, as it does not appear in the groupByFields.",
            )

    def execute(self, database, filters):  # aggregated.cpp:58-66,90-149
        if not self.group_by_fields:
            count = 0
            for bitmap in filters:
                count = (count + card(bitmap)) & UINT32
            return [{"count": _to_int32(count)}]
        for name in self.group_by_fields:
            check_silo_query(database.column_type(name) is not None, "Metadata field '" + name + "' to group by not found")
        # tuples are keyed by their raw column values (tuple.cpp:29-80); the row order of the result is the iteration
        # order of an unordered_map, i.e. unspecified: callers compare as multisets unless orderByFields fixes it
        counts = {}
        for partition, bitmap in zip(database.partitions, filters):
            for row in ids_from_bits(bitmap):
                key = tuple(raw_tuple_value(database, partition, name, row) for name in self.group_by_fields)
                counts[key] = counts.get(key, 0) + 1
        out = []
        for key, count in counts.items():
            fields = {name: json_tuple_value(database, name, value) for name, value in zip(self.group_by_fields, key)}
            fields["count"] = _to_int32(count)
            out.append(fields)
        return out


def raw_tuple_value(database, partition, name, row):
    """What assignTupleField stores for a row (tuple.cpp:29-80), with dictionary ids replaced by their strings and a
    NaN made comparable (bytewise tuple equality: every NULL float is the same std::nan(""))."""
    kind = database.column_type(name)
    if kind == "string":
        return partition.string_columns[name][row]
    if kind == "indexed_string":
        return partition.indexed_string_columns[name][row]
    if kind == "pango_lineage":
        return partition.pango_lineage_columns[name].lookup_aliased_value(row)
    if kind == "int":
        return partition.int_columns[name][row]
    if kind == "float":
        value = partition.float_columns[name][row]
        return "NaN" if value != value else value
    if kind == "date":
        return partition.date_columns[name][row]
    if kind in ("insertion", "aa_insertion"):
        return partition.insertion_columns[name][row]
    raise KeyError(name)


def json_tuple_value(database, name, raw):
    """tupleFieldToValueType (tuple.cpp:82-160): NULLs ("" / INT32_MIN / NaN / date 0) become JSON null."""
    kind = database.column_type(name)
    if kind == "date":
        return date_to_string(raw)
    if kind == "int":
        return None if raw == INT32_MIN else raw
    if kind == "float":
        return None if raw == "NaN" else raw
    return None if raw == "" else raw


def tuple_compare(database, fields, order_by_fields):
    """Tuple::compareLess (tuple.cpp:372-387) on raw values: dates / ints numerically (NULL = smallest), floats with
    NaN LAST (compareDouble :162-182), strings bytewise (the NULL "" first)."""
    import functools

    def compare_values(kind, a, b):
        if kind == "float":
            a_nan, b_nan = a == "NaN", b == "NaN"
            if a_nan or b_nan:
                return 0 if (a_nan and b_nan) else (1 if a_nan else -1)
        if isinstance(a, str):
            a, b = a.encode(), b.encode()
        return -1 if a < b else (1 if a > b else 0)

    def compare(row_a, row_b):
        for f in order_by_fields:
            index = fields.index(f.name)
            c = compare_values(database.column_type(f.name), row_a[index], row_b[index])
            if c != 0:
                return c if f.ascending else -c
        return 0

    return functools.cmp_to_key(compare)


class Details(Action):  # details.cpp
    def __init__(self, fields):
        super().__init__()
        self.fields = fields

    def field_list(self, database):  # parseFields :22-35
        if not self.fields:
            return [name for name, _ in database.metadata]
        for name in self.fields:
            check_silo_query(database.column_type(name) is not None, "Metadata field " + name + " not found.")
        return list(self.fields)

    def validate_order_by_fields(self, database):  # :43-59
        fields = self.field_list(database)
        for f in self.order_by_fields:
            check_silo_query(f.name in fields, "OrderByField " + f.name + " is not contained in the result of this operation.")

    def execute_and_order(self, database, filters):  # :186-219
        self.validate_order_by_fields(database)
        fields = self.field_list(database)
        tuples = []
        for partition, bitmap in zip(database.partitions, filters):
            for row in ids_from_bits(bitmap):
                tuples.append(tuple(raw_tuple_value(database, partition, name, row) for name in fields))
        if self.order_by_fields:
            tuples.sort(key=tuple_compare(database, fields, self.order_by_fields))
        if self.limit is not None:
            tuples = tuples[: self.limit + (self.offset or 0)]
        result = [{name: json_tuple_value(database, name, value) for name, value in zip(fields, row)} for row in tuples]
        return self.apply_offset_and_limit(result)


class FastaAligned(Action):  # fasta_aligned.cpp
    def __init__(self, sequence_names):
        super().__init__()
        self.sequence_names = sequence_names

    def validate_order_by_fields(self, database):  # :28-42
        for f in self.order_by_fields:
            check_silo_query(
                f.name == database.primary_key or f.name in self.sequence_names,
                "The only fields returned by the FastaAligned action are " + ",".join(self.sequence_names) + " and " + database.primary_key,
            )

    def execute(self, database, filters):  # :85-136; reconstructSequence :44-83 = the stored symbol of every cell
        for name in self.sequence_names:
            check_silo_query(
                name in database.nuc_references or name in database.aa_references,
                "Database does not contain a sequence with name: '" + name + "'",
            )
        total = sum(card(bitmap) for bitmap in filters)
        check_silo_query(total < 10001, "FastaAligned action currently limited to 10000 sequences")
        out = []
        for partition, bitmap in zip(database.partitions, filters):
            for row in ids_from_bits(bitmap):
                entry = {database.primary_key: json_tuple_value(
                    database, database.primary_key, raw_tuple_value(database, partition, database.primary_key, row))}
                for name in self.sequence_names:
                    is_nuc = name in database.nuc_references
                    store = (partition.nuc_sequences if is_nuc else partition.aa_sequences)[name]
                    alphabet = Nucleotide if is_nuc else AminoAcid
                    entry[name] = "".join(alphabet.symbol_to_char(store.symbol_at(position, row)) for position in range(len(store.positions)))
                out.append(entry)
        return out


def _to_int32(value):
    value &= UINT32
    return value - (1 << 32) if value & 0x80000000 else value


class Mutations(Action):
    def __init__(self, alphabet, sequence_names, min_proportion):
        super().__init__()
        self.alphabet = alphabet
        self.sequence_names = sequence_names
        self.min_proportion = min_proportion

    def validate_order_by_fields(self, database):  # mutations.cpp:166-182
        for f in self.order_by_fields:
            check_silo_query(
                f.name in ("mutation", "proportion", "count"),
                "OrderByField " + f.name + " is not contained in the result of this operation.",
            )

    def stores(self, database, partition):
        return partition.nuc_sequences if self.alphabet is Nucleotide else partition.aa_sequences

    def pre_filter(self, database, filters):  # mutations.cpp:35-62
        to_evaluate = {}
        for partition, bitmap in zip(database.partitions, filters):
            cardinality = card(bitmap)
            if cardinality == 0:
                continue
            kind = "full" if cardinality == partition.sequence_count else "mixed"
            for name, store in self.stores(database, partition).items():
                to_evaluate.setdefault(name, {"mixed": [], "full": []})[kind].append((bitmap, store))
        return to_evaluate

    def counts_per_position(self, sequence_length, prefiltered):  # mutations.cpp:64-164
        symbols = self.alphabet.SYMBOLS
        counts = {s: [0] * sequence_length for s in symbols}
        for pos in range(sequence_length):
            for bitmap, store in prefiltered["mixed"]:  # :64-96
                position = store.positions[pos]
                filter_ids = None
                for symbol in symbols:
                    if position.deleted == symbol:
                        counts[symbol][pos] = (counts[symbol][pos] + card(bitmap)) & UINT32
                        if filter_ids is None:
                            filter_ids = ids_from_bits(bitmap)
                        for idx in filter_ids:
                            if pos in store.missing_symbol_bitmaps[idx]:
                                counts[symbol][pos] = (counts[symbol][pos] - 1) & UINT32
                        continue
                    column = position.bitmaps[symbol]
                    symbol_count = card(bitmap & ~column) if position.flipped == symbol else card(bitmap & column)
                    counts[symbol][pos] = (counts[symbol][pos] + symbol_count) & UINT32
                    if position.deleted is not None and symbol != position.deleted:
                        counts[position.deleted][pos] = (counts[position.deleted][pos] - symbol_count) & UINT32
            for bitmap, store in prefiltered["full"]:  # :98-136
                position = store.positions[pos]
                for symbol in symbols:
                    if position.deleted == symbol:
                        counts[symbol][pos] = (counts[symbol][pos] + store.sequence_count) & UINT32
                        for row in store.missing_symbol_bitmaps:
                            if pos in row:
                                counts[symbol][pos] = (counts[symbol][pos] - 1) & UINT32
                        continue
                    cardinality = card(position.bitmaps[symbol])
                    symbol_count = store.sequence_count - cardinality if position.flipped == symbol else cardinality
                    counts[symbol][pos] = (counts[symbol][pos] + symbol_count) & UINT32
                    if position.deleted is not None and symbol != position.deleted:
                        counts[position.deleted][pos] = (counts[position.deleted][pos] - cardinality) & UINT32
        return counts

    def add_mutations_to_output(self, sequence_name, reference, prefiltered, output):  # mutations.cpp:184-232
        sequence_length = len(reference)
        counts = self.counts_per_position(sequence_length, prefiltered)
        valid = self.alphabet.VALID_MUTATION_SYMBOLS
        for pos in range(sequence_length):
            total = 0
            for symbol in valid:
                total = (total + counts[symbol][pos]) & UINT32
            if total == 0:
                continue
            if self.min_proportion == 0:
                threshold_count = 0
            else:
                threshold_count = int(math.ceil(float(total) * self.min_proportion) - 1) & UINT32
            ref_symbol = reference[pos]
            for symbol in valid:
                if symbol != ref_symbol:
                    count = counts[symbol][pos]
                    if count > threshold_count:
                        output.append({
                            "mutation": self.alphabet.symbol_to_char(ref_symbol) + str(pos + 1) + self.alphabet.symbol_to_char(symbol),
                            "sequenceName": sequence_name,
                            "proportion": float(count) / float(total),
                            "count": _to_int32(count),
                        })

    def execute(self, database, filters):  # mutations.cpp:234-272
        references = database.sequence_store_names(self.alphabet)
        names = []
        for name in self.sequence_names:
            check_silo_query(
                name in references,
                "Database does not contain the " + self.alphabet.NAME_LOWER + " sequence with name: '" + name + "'",
            )
            names.append(name)
        if not self.sequence_names:
            names = sorted(references)  # std::map iteration order
        to_evaluate = self.pre_filter(database, filters)
        output = []
        for name in names:
            if name in to_evaluate:
                self.add_mutations_to_output(name, references[name], to_evaluate[name], output)
        return output


class Fasta(Action):  # fasta.cpp: primary key + unaligned nucleotide sequences of the selected rows
    SEQUENCE_LIMIT = 10000

    def __init__(self, sequence_names):
        super().__init__()
        self.sequence_names = sequence_names

    def validate_order_by_fields(self, database):  # :42-57
        for f in self.order_by_fields:
            check_silo_query(
                f.name == database.primary_key or f.name in self.sequence_names,
                "The only fields returned by the Fasta action are " + ",".join(self.sequence_names) + " and " + database.primary_key,
            )

    def execute(self, database, filters):  # :214-245; every nucleotide sequence has an unaligned store (database.cpp:664-673)
        for name in self.sequence_names:
            check_silo_query(name in database.nuc_references, "Database does not contain an unaligned sequence with name: '" + name + "'")
        total = sum(card(bitmap) for bitmap in filters)
        check_silo_query(total <= self.SEQUENCE_LIMIT, "Fasta action currently limited to " + str(self.SEQUENCE_LIMIT) + " sequences")
        out = []
        for partition, bitmap in zip(database.partitions, filters):
            for row in ids_from_bits(bitmap):
                entry = {database.primary_key: json_tuple_value(
                    database, database.primary_key, raw_tuple_value(database, partition, database.primary_key, row))}
                if entry[database.primary_key] is None:
                    raise RuntimeError("Detected primary_key in column '" + database.primary_key + "' that is null.")
                for name in self.sequence_names:
                    sequences = partition.unaligned_nuc_sequences.get(name)
                    entry[name] = sequences[row] if sequences is not None else None
                out.append(entry)
        return out


class InsertionAggregation(Action):  # insertions.cpp
    def __init__(self, alphabet, column_names, sequence_names):
        super().__init__()
        self.alphabet, self.column_names, self.sequence_names = alphabet, column_names, sequence_names

    def validate_order_by_fields(self, database):  # :41-59
        for f in self.order_by_fields:
            check_silo_query(
                f.name in ("position", "insertions", "sequenceName", "count"),
                "OrderByField " + f.name + " is not contained in the result of this operation.",
            )

    def execute(self, database, filters):  # :126-258; rows come out of unordered_maps: the order is unspecified
        symbol_name = "Nucleotide" if self.alphabet is Nucleotide else "Amino Acid"  # SymbolType::SYMBOL_NAME
        wanted_kind = "insertion" if self.alphabet is Nucleotide else "aa_insertion"
        for column_name in self.column_names:
            check_silo_query(
                database.column_type(column_name) == wanted_kind,
                "The database does not contain the " + symbol_name + " column '" + column_name + "'",
            )
        references = database.sequence_store_names(self.alphabet)
        for sequence_name in self.sequence_names:
            check_silo_query(
                sequence_name in references, "The database does not contain the " + symbol_name + " sequence '" + sequence_name + "'"
            )
        counts = {}  # sequence name -> (position, insertion) -> count
        for partition, bitmap in zip(database.partitions, filters):
            indexes = partition.nuc_insertion_indexes if self.alphabet is Nucleotide else partition.aa_insertion_indexes
            if card(bitmap) == 0:
                continue
            selected = set(ids_from_bits(bitmap))
            for column_name, index in indexes.items():
                if self.column_names and column_name not in self.column_names:
                    continue
                for sequence_name, positions in index.items():
                    if self.sequence_names and sequence_name not in self.sequence_names:
                        continue
                    per_sequence = counts.setdefault(sequence_name, {})
                    for position, insertions in positions.items():
                        for insertion, row_set in insertions.items():
                            count = len(row_set & selected)
                            if count > 0:
                                per_sequence[(position, insertion)] = per_sequence.get((position, insertion), 0) + count
        out = []
        for sequence_name, per_sequence in counts.items():
            for (position, insertion), count in per_sequence.items():
                out.append({"position": position, "sequenceName": sequence_name, "insertions": insertion, "count": _to_int32(count)})
        return out


def parse_insertions(node, alphabet):  # insertions.cpp:260-302
    check_silo_query(
        "sequenceName" not in node or isinstance(node["sequenceName"], (str, list)),
        "Insertions action can have the field sequenceName of type string or an array of strings, but no other type",
    )
    sequence_names = []
    if isinstance(node.get("sequenceName"), list):
        for child in node["sequenceName"]:
            check_silo_query(
                isinstance(child, str),
                "The field sequenceName of the Insertions action must have type string or an array, if present. Found:"
                + json.dumps(child, separators=(",", ":")),
            )
            sequence_names.append(child)
    elif isinstance(node.get("sequenceName"), str):
        sequence_names.append(node["sequenceName"])
    check_silo_query(
        "column" not in node or isinstance(node["column"], (str, list)),
        "Insertions action can have the field column of type string or an array of strings, but no other type",
    )
    column_names = []
    if isinstance(node.get("column"), list):
        for child in node["column"]:
            check_silo_query(
                isinstance(child, str),
                "The field column of the Insertions action must have type string or an array, if present. Found:"
                + json.dumps(child, separators=(",", ":")),
            )
            column_names.append(child)
    elif isinstance(node.get("column"), str):
        column_names.append(node["column"])
    return InsertionAggregation(alphabet, column_names, sequence_names)


def parse_order_by_field(node):  # action.cpp:119-142
    if isinstance(node, str):
        return OrderByField(node, True)
    message = (
        "The orderByField '" + json.dumps(node, separators=(",", ":")) + "' must be either a string or an object containing "
        "the fields 'field':string and 'order':string, where the value of order is 'ascending' or 'descending'"
    )
    check_silo_query(
        isinstance(node, dict) and isinstance(node.get("field"), str) and isinstance(node.get("order"), str), message
    )
    check_silo_query(node["order"] in ("ascending", "descending"), message)
    return OrderByField(node["field"], node["order"] == "ascending")


def parse_mutations(node, alphabet):  # mutations.cpp:274-316
    check_silo_query(
        "sequenceName" not in node or isinstance(node["sequenceName"], (str, list)),
        "Mutations action can have the field sequenceName of type string or an array of strings, but no other type",
    )
    names = []
    if isinstance(node.get("sequenceName"), list):
        for child in node["sequenceName"]:
            check_silo_query(
                isinstance(child, str),
                "The field sequenceName of Mutations action must have type string or an array, if present. Found:"
                + json.dumps(child, separators=(",", ":")),
            )
            names.append(child)
    elif isinstance(node.get("sequenceName"), str):
        names.append(node["sequenceName"])
    check_silo_query(
        isinstance(node.get("minProportion"), (int, float)) and not isinstance(node.get("minProportion"), bool),
        "Mutations action must contain the field minProportion of type number with limits [0.0, 1.0]. Only mutations are "
        "returned if the proportion of sequences having this mutation, is at least minProportion",
    )
    min_proportion = float(node["minProportion"])
    if min_proportion < 0 or min_proportion > 1:
        raise QueryParseException("Invalid proportion: minProportion must be in interval [0.0, 1.0]")
    return Mutations(alphabet, names, min_proportion)


def parse_action(node):  # action.cpp:144-187
    check_silo_query("type" in node, "The field 'type' is required in any action")
    check_silo_query(
        isinstance(node["type"], str),
        "The field 'type' in all actions needs to be a string, but is: " + json.dumps(node["type"]),
    )
    kind = node["type"]
    if kind == "Aggregated":
        action = Aggregated(list(node.get("groupByFields", [])))
    elif kind == "Mutations":
        action = parse_mutations(node, Nucleotide)
    elif kind == "AminoAcidMutations":
        action = parse_mutations(node, AminoAcid)
    elif kind == "Details":  # details.cpp:221-224
        action = Details(list(node.get("fields", [])))
    elif kind == "FastaAligned":  # fasta_aligned.cpp:138-161
        check_silo_query(
            isinstance(node.get("sequenceName"), (str, list)),
            "FastaAligned action must have the field sequenceName of type string or an array of strings",
        )
        names = []
        if isinstance(node["sequenceName"], list):
            for child in node["sequenceName"]:
                check_silo_query(
                    isinstance(child, str),
                    "FastaAligned action must have the field sequenceName of type string or an array of strings; while parsing "
                    "array encountered the element " + json.dumps(child, separators=(",", ":")) + " which is not of type string",
                )
                names.append(child)
        else:
            names.append(node["sequenceName"])
        action = FastaAligned(names)
    elif kind == "Insertions":
        action = parse_insertions(node, Nucleotide)
    elif kind == "AminoAcidInsertions":
        action = parse_insertions(node, AminoAcid)
    elif kind == "Fasta":  # fasta.cpp:247-270
        check_silo_query(
            isinstance(node.get("sequenceName"), (str, list)),
            "Fasta action must have the field sequenceName of type string or an array of strings",
        )
        names = []
        if isinstance(node["sequenceName"], list):
            for child in node["sequenceName"]:
                check_silo_query(
                    isinstance(child, str),
                    "Fasta action must have the field sequenceName of type string or an array of strings; while parsing array "
                    "encountered the element " + json.dumps(child, separators=(",", ":")) + " which is not of type string",
                )
                names.append(child)
        else:
            names.append(node["sequenceName"])
        action = Fasta(names)
    else:
        raise QueryParseException(kind + " is not a valid action")
    order_by = [parse_order_by_field(f) for f in node.get("orderByFields", [])]
    check_silo_query("limit" not in node or _is_unsigned(node["limit"]), "If the action contains a limit, it must be a non-negative number")
    check_silo_query("offset" not in node or _is_unsigned(node["offset"]), "If the action contains an offset, it must be a non-negative number")
    action.order_by_fields = order_by
    action.limit = node.get("limit")
    action.offset = node.get("offset")
    return action


# ------------------------------------------------------------------------------------------------
# engine (query.cpp:13-28, query_engine.cpp:30-68)
# ------------------------------------------------------------------------------------------------
def parse_query(query):
    if isinstance(query, str):
        try:
            query = json.loads(query)
        except json.JSONDecodeError as error:
            raise QueryParseException("The query was not a valid JSON: " + str(error))
    if (
        not isinstance(query, dict)
        or not isinstance(query.get("filterExpression"), dict)
        or not isinstance(query.get("action"), dict)
    ):
        raise QueryParseException("Query json must contain filterExpression and action.")
    return parse_expression(query["filterExpression"]), parse_action(query["action"])


def evaluate_filters(database, expression):
    """Per-partition filter bitsets (query_engine.cpp:40-49)."""
    return [expression.compile(database, partition, NONE).evaluate() for partition in database.partitions]


def execute_query(database, query):
    """Returns the list that the reference serialises as {"queryResult": [...]} (query_result.cpp:10-25)."""
    expression, action = parse_query(query)
    filters = evaluate_filters(database, expression)
    return action.execute_and_order(database, filters)
