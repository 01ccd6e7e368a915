// filter_expressions.cpp — logical filter tree: JSON -> Expression -> compile() -> operators.
// Mirrors src/silo/query_engine/filter_expressions/*.cpp of the reference (cited per function),
// including the algebraic rewrites and the observable quirks listed in SURVEY.md §8(a).
#include <algorithm>
#include <iterator>

#include <cmath>
#include <regex>
#include <type_traits>
#include <climits>

#include "query_engine.h"

namespace silo::query_engine::filter_expressions {

using operators::Operator;
using operators::OperatorVector;

namespace {

RowSpace rowsOf(const DatabasePartition& partition) {
   return {partition.sequence_count, &partition};
}

// nucleotide_symbol_equals.cpp:28-73
using NS = Nucleotide::Symbol;
const std::array<std::vector<NS>, Nucleotide::COUNT> AMBIGUITY_NUC_SYMBOLS{{
   {NS::GAP},
   {NS::A, NS::R, NS::M, NS::W, NS::D, NS::H, NS::V, NS::N},
   {NS::C, NS::Y, NS::M, NS::S, NS::B, NS::H, NS::V, NS::N},
   {NS::G, NS::R, NS::K, NS::S, NS::B, NS::D, NS::V, NS::N},
   {NS::T, NS::Y, NS::K, NS::W, NS::B, NS::D, NS::H, NS::N},
   {NS::R}, {NS::Y}, {NS::S}, {NS::W}, {NS::K}, {NS::M}, {NS::B}, {NS::D}, {NS::H}, {NS::V}, {NS::N},
}};

/// Leaf operator for "row has `symbol` at `position`".  The reference distinguishes flipped and deleted
/// storage here (nucleotide_symbol_equals.cpp:144-188); those are storage tricks with no observable
/// effect: the dense store holds the true membership plane of every symbol.
template <typename SymbolType>
std::unique_ptr<Operator> symbolPlane(
   const Database& database, const SequenceStorePartition<SymbolType>& store, const DatabasePartition& partition, uint32_t position,
   typename SymbolType::Symbol symbol
) {
   const RowSpace rows = rowsOf(partition);
   const bool exchange = database.shard_by_position && database.shard_world > 1 && database.broadcast != nullptr;
   if (exchange) {
      // Position-range sharding keeps only a slice of the genome per rank, but a filter leaf may sit at any
      // position: the rank that owns it broadcasts the plane (Wp words), every rank runs this same code in
      // the same order (SPMD), so the collectives line up.
      const size_t row_bytes = static_cast<size_t>(partition.rowWords()) * sizeof(uint64_t);
      const uint32_t owner = database.ownerOfPosition(position, store.reference_sequence.size());
      auto buffer = std::make_shared<DeviceBuffer>();
      void* payload = nullptr;
      if (owner == database.shard_rank) {
         const uint64_t* plane = store.getBitmap(position, symbol);
         if (plane == nullptr) {  // sparsely stored symbol: materialise it first
            *buffer = partition.pool.acquire(row_bytes);
            checkGpu(
               silo_gpu_store_sparse_plane(
                  store.store, store.seqstore_id, position - store.position_begin, static_cast<uint32_t>(symbol), buffer->as<uint64_t>(),
                  queryStream()
               ),
               "silo_gpu_store_sparse_plane"
            );
            plane = buffer->as<uint64_t>();
         }
         payload = const_cast<uint64_t*>(plane);  // the root only sends
      } else {
         *buffer = partition.pool.acquire(row_bytes);
         payload = buffer->get();
      }
      const int status = database.broadcast(database.broadcast_context, payload, row_bytes, owner, queryStream());
      if (status != 0) {
         throw DeviceException("broadcast of a filter leaf failed with status " + std::to_string(status));
      }
      auto scan = std::make_unique<operators::IndexScan>(static_cast<const uint64_t*>(payload), rows);
      scan->received = std::move(buffer);
      return scan;
   }
   if (!store.holds(position)) {
      throw std::runtime_error(
         "position " + std::to_string(position + 1) + " is not resident on this rank (position-range shard " +
         std::to_string(store.position_begin + 1) + ".." + std::to_string(store.position_end) +
         ") and no broadcast callback is installed"
      );
   }
   const uint64_t* plane = store.getBitmap(position, symbol);
   if (symbol == SymbolType::SYMBOL_MISSING) {
      // nucleotide_symbol_equals.cpp:131-143 / aa_symbol_equals.cpp:55-62
      if (plane == nullptr) {  // the store keeps the missing symbol as runs: the position's plane is built when the operator is lowered
         return std::make_unique<operators::BitmapSelection>(
            store.seqstore_id, position - store.position_begin, static_cast<uint32_t>(symbol), rows, operators::BitmapSelection::CONTAINS, position
         );
      }
      return std::make_unique<operators::BitmapSelection>(plane, rows, operators::BitmapSelection::CONTAINS, position);
   }
   if (plane == nullptr) {
      return std::make_unique<operators::IndexScan>(store.seqstore_id, position - store.position_begin, static_cast<uint32_t>(symbol), rows);
   }
   return std::make_unique<operators::IndexScan>(plane, rows);
}

}  // namespace

Expression::AmbiguityMode invertMode(Expression::AmbiguityMode mode) {  // expression.cpp:38-46
   if (mode == Expression::UPPER_BOUND) {
      return Expression::LOWER_BOUND;
   }
   if (mode == Expression::LOWER_BOUND) {
      return Expression::UPPER_BOUND;
   }
   return mode;
}

// ---- True / False ---------------------------------------------------------------------------------
std::string True::toString(const Database& /*database*/) const {
   return "True";
}
std::unique_ptr<Operator> True::compile(const Database& /*database*/, const DatabasePartition& database_partition, AmbiguityMode /*mode*/) const {
   return std::make_unique<operators::Full>(rowsOf(database_partition));
}
std::string False::toString(const Database& /*database*/) const {
   return "False";
}
std::unique_ptr<Operator> False::compile(const Database& /*database*/, const DatabasePartition& database_partition, AmbiguityMode /*mode*/) const {
   return std::make_unique<operators::Empty>(rowsOf(database_partition));
}

// ---- Boolean combinators: And, Or, N-Of -------------------------------------------------------------
// The reference rewrites each of the three with its own case analysis (and.cpp:101-227, or.cpp:41-94, nof.cpp:35-280).
// Here all three go through ONE normal form and four constructors.  The compiled children of a combinator are sorted
// into a Literals record; "every literal", "some literal", "no literal" and "k of the literals" are then built from
// it.  The sets that come out are the reference's (the e2e goldens, the operator vectors and the randomised trees
// against the oracle pin that); the operator trees need not be, and the device lowers whatever tree it gets into one
// bit-program anyway.
namespace {

/// The compiled children of a combinator.  A child that compiled to a Complement is kept WITHOUT its complement in
/// `inverted`; children that hold for every row are only counted, children that hold for no row only flagged.
struct Literals {
   OperatorVector plain;
   OperatorVector inverted;
   std::vector<operators::Predicate> predicates;  // metadata comparisons lifted out of Selection children (And only)
   int always = 0;
   bool never = false;

   [[nodiscard]] int size() const { return static_cast<int>(plain.size() + inverted.size()); }
};

enum class Flatten { CONJUNCTIONS, DISJUNCTIONS, NOTHING };

/// Files one compiled child.  An And splices the literals of a nested Intersection and lifts the predicates of a
/// nested Selection (and.cpp:124-151); an Or splices a nested Union (or.cpp:56-62); an N-Of keeps children whole.
void file(Literals& literals, std::unique_ptr<Operator> child, Flatten flatten) {
   switch (child->type()) {
      case operators::FULL:
         ++literals.always;
         return;
      case operators::EMPTY:
         literals.never = true;
         return;
      case operators::COMPLEMENT:
         literals.inverted.push_back(child->negate());
         return;
      case operators::INTERSECTION:
         if (flatten == Flatten::CONJUNCTIONS) {
            auto& nested = static_cast<operators::Intersection&>(*child);
            std::move(nested.children.begin(), nested.children.end(), std::back_inserter(literals.plain));
            std::move(nested.negated_children.begin(), nested.negated_children.end(), std::back_inserter(literals.inverted));
            return;
         }
         break;
      case operators::UNION:
         if (flatten == Flatten::DISJUNCTIONS) {
            auto& nested = static_cast<operators::Union&>(*child);
            std::move(nested.children.begin(), nested.children.end(), std::back_inserter(literals.plain));
            return;
         }
         break;
      case operators::SELECTION:
         if (flatten == Flatten::CONJUNCTIONS) {
            auto& nested = static_cast<operators::Selection&>(*child);
            literals.predicates.insert(literals.predicates.end(), nested.predicates.begin(), nested.predicates.end());
            if (nested.child != nullptr) {
               file(literals, std::move(nested.child), flatten);
            }
            return;
         }
         break;
      default:
         break;
   }
   literals.plain.push_back(std::move(child));
}

Literals compileLiterals(
   const ExpressionVector& children, const Database& database, const DatabasePartition& partition, Expression::AmbiguityMode mode, Flatten flatten
) {
   Literals literals;
   for (const auto& child : children) {
      file(literals, child->compile(database, partition, mode), flatten);
   }
   return literals;
}

/// Rows where every plain literal holds and no inverted one does.
std::unique_ptr<Operator> everyLiteral(OperatorVector&& plain, OperatorVector&& inverted, RowSpace rows) {
   if (plain.empty()) {
      if (inverted.empty()) {
         return std::make_unique<operators::Full>(rows);
      }
      std::unique_ptr<Operator> excluded =
         inverted.size() == 1 ? std::move(inverted.front()) : std::make_unique<operators::Union>(std::move(inverted), rows);
      return std::make_unique<operators::Complement>(std::move(excluded), rows);
   }
   if (plain.size() == 1 && inverted.empty()) {
      return std::move(plain.front());
   }
   return std::make_unique<operators::Intersection>(std::move(plain), std::move(inverted), rows);
}

/// Rows where some plain literal holds or some inverted one does not: De Morgan of everyLiteral with the roles swapped.
std::unique_ptr<Operator> someLiteral(OperatorVector&& plain, OperatorVector&& inverted, RowSpace rows) {
   if (inverted.empty()) {
      if (plain.empty()) {
         return std::make_unique<operators::Empty>(rows);
      }
      return plain.size() == 1 ? std::move(plain.front()) : std::make_unique<operators::Union>(std::move(plain), rows);
   }
   return std::make_unique<operators::Complement>(everyLiteral(std::move(inverted), std::move(plain), rows), rows);
}

/// Rows where at least (or exactly) `wanted` of the literals hold; literals that always hold have been taken off
/// `wanted` by the caller, literals that never hold are gone.
std::unique_ptr<Operator> countLiterals(Literals&& literals, int wanted, bool exactly, RowSpace rows) {
   const int available = literals.size();
   if (wanted > available || (exactly && wanted < 0)) {
      return std::make_unique<operators::Empty>(rows);
   }
   if (wanted <= 0 && !exactly) {
      return std::make_unique<operators::Full>(rows);
   }
   if (wanted == available) {  // all of them (for "exactly 0 of 0": every row)
      return everyLiteral(std::move(literals.plain), std::move(literals.inverted), rows);
   }
   if (wanted == 0) {  // exactly none
      return everyLiteral(std::move(literals.inverted), std::move(literals.plain), rows);
   }
   if (wanted == 1 && !exactly) {
      return someLiteral(std::move(literals.plain), std::move(literals.inverted), rows);
   }
   return std::make_unique<operators::Threshold>(std::move(literals.plain), std::move(literals.inverted), static_cast<uint32_t>(wanted), exactly, rows);
}

std::string describe(const ExpressionVector& children, const Database& database, const char* separator) {
   std::string text;
   for (const auto& child : children) {
      text += (text.empty() ? "" : separator) + child->toString(database);
   }
   return text;
}

}  // namespace

std::string And::toString(const Database& database) const {
   return "And(" + describe(children, database, " & ") + ")";
}

std::unique_ptr<Operator> And::compile(const Database& database, const DatabasePartition& database_partition, AmbiguityMode mode) const {
   const RowSpace rows = rowsOf(database_partition);
   Literals literals = compileLiterals(children, database, database_partition, mode, Flatten::CONJUNCTIONS);
   if (literals.never) {
      return std::make_unique<operators::Empty>(rows);
   }
   if (literals.predicates.empty()) {
      return everyLiteral(std::move(literals.plain), std::move(literals.inverted), rows);
   }
   if (literals.size() == 0) {
      return std::make_unique<operators::Selection>(std::move(literals.predicates), rows);
   }
   // the comparisons filter what the index arithmetic leaves (and.cpp:203-227)
   return std::make_unique<operators::Selection>(
      everyLiteral(std::move(literals.plain), std::move(literals.inverted), rows), std::move(literals.predicates), rows
   );
}

std::string Or::toString(const Database& database) const {
   return "Or(" + describe(children, database, " | ") + ")";
}

std::unique_ptr<Operator> Or::compile(const Database& database, const DatabasePartition& database_partition, AmbiguityMode mode) const {
   const RowSpace rows = rowsOf(database_partition);
   Literals literals = compileLiterals(children, database, database_partition, mode, Flatten::DISJUNCTIONS);
   if (literals.always > 0) {
      return std::make_unique<operators::Full>(rows);
   }
   return someLiteral(std::move(literals.plain), std::move(literals.inverted), rows);
}

// ---- Negation / Maybe / Exact ----------------------------------------------------------------------
std::string Negation::toString(const Database& database) const {
   return "!(" + child->toString(database) + ")";
}
std::unique_ptr<Operator> Negation::compile(const Database& database, const DatabasePartition& database_partition, AmbiguityMode mode) const {
   // an upper bound of the child is a lower bound of its complement: negation.cpp:27-34
   return child->compile(database, database_partition, invertMode(mode))->negate();
}
std::string Maybe::toString(const Database& database) const {
   return "Maybe (" + child->toString(database) + ")";
}
std::unique_ptr<Operator> Maybe::compile(const Database& database, const DatabasePartition& database_partition, AmbiguityMode /*mode*/) const {
   return child->compile(database, database_partition, AmbiguityMode::UPPER_BOUND);  // maybe.cpp:26-32
}
std::string Exact::toString(const Database& database) const {
   return "Exact ( " + child->toString(database) + ")";
}
std::unique_ptr<Operator> Exact::compile(const Database& database, const DatabasePartition& database_partition, AmbiguityMode /*mode*/) const {
   return child->compile(database, database_partition, AmbiguityMode::LOWER_BOUND);  // exact.cpp:26-32
}

// ---- N-Of ------------------------------------------------------------------------------------------
std::string NOf::toString(const Database& database) const {
   std::string text = match_exactly ? "[exactly-" : "[";
   text += std::to_string(number_of_matchers) + "-of:";
   for (const auto& child : children) {
      text += child->toString(database) + ", ";
   }
   return text + "]";
}

std::unique_ptr<Operator> NOf::compile(const Database& database, const DatabasePartition& database_partition, AmbiguityMode mode) const {
   const RowSpace rows = rowsOf(database_partition);
   const auto atLeastOrExactly = [&](int wanted, bool exactly) {
      Literals literals = compileLiterals(children, database, database_partition, mode, Flatten::NOTHING);
      // a literal that holds for every row is one match already (nof.cpp:202-203)
      return countLiterals(std::move(literals), wanted - literals.always, exactly, rows);
   };
   // "exactly k" has no monotone bound of its own: under Maybe / Exact it is "at least k" without "at least k + 1", both
   // bounded the same way (nof.cpp:220-258, 268-271)
   if (match_exactly && mode != NONE && number_of_matchers < static_cast<int>(children.size())) {
      OperatorVector reached;
      OperatorVector exceeded;
      reached.push_back(atLeastOrExactly(number_of_matchers, false));
      exceeded.push_back(atLeastOrExactly(number_of_matchers + 1, false));
      return std::make_unique<operators::Intersection>(std::move(reached), std::move(exceeded), rows);
   }
   return atLeastOrExactly(number_of_matchers, match_exactly);
}

namespace {

/// "Symbol or any ambiguity code that may stand for it" (UPPER_BOUND, nucleotide_symbol_equals.cpp:137-149) is the
/// OR of up to 8 planes, most of them nearly empty IUPAC planes.  The database is immutable, so the combined plane
/// is built once (one fused launch) and kept in the partition's derived-plane cache next to the materialised sparse
/// planes; later queries read ONE column instead of up to eight.  Beyond the cache budget the plain Or is returned.
template <typename Expand>
std::unique_ptr<Operator> cachedUpperBoundPlane(
   const Database& database, const DatabasePartition& partition, uint32_t seqstore_id, uint32_t position, uint32_t symbol, Expand&& expand
) {
   const RowSpace rows = rowsOf(partition);
   if (database.broadcast != nullptr) {
      return expand();  // leaf exchange between ranks is in play: every leaf travels on its own
   }
   const uint64_t key = (uint64_t{1} << 63) | (static_cast<uint64_t>(seqstore_id) << 40) | (static_cast<uint64_t>(position) << 8) | symbol;
   const size_t row_bytes = static_cast<size_t>(partition.rowWords()) * sizeof(uint64_t);
   {
      const std::shared_lock<std::shared_mutex> lock(partition.sparse_cache_mutex);
      const auto found = partition.sparse_cache.find(key);
      if (found != partition.sparse_cache.end()) {  // the usual case: no expansion is even built
         return std::make_unique<operators::IndexScan>(found->second.as<uint64_t>(), rows);
      }
   }
   std::unique_ptr<Operator> expanded = expand();
   if (expanded->type() == operators::INDEX_SCAN || expanded->type() == operators::EMPTY || expanded->type() == operators::FULL) {
      return expanded;  // there is nothing to combine
   }
   {
      const std::shared_lock<std::shared_mutex> lock(partition.sparse_cache_mutex);
      if ((partition.sparse_cache.size() + 1) * row_bytes > DatabasePartition::SPARSE_CACHE_BYTES) {
         return expanded;
      }
   }
   DeviceBuffer buffer = partition.pool.acquire(row_bytes);
   {
      ProgramBuilder builder(rows);
      const uint32_t slot = expanded->lower(builder);
      builder.run(slot, buffer.as<uint64_t>(), nullptr, queryStream());
   }
   // other threads (on their own streams) may pick the plane up from the cache at once: finish it first
   checkGpu(silo_gpu_stream_synchronize(queryStream()), "silo_gpu_stream_synchronize");
   const uint64_t* pointer = buffer.as<uint64_t>();
   {
      const std::unique_lock<std::shared_mutex> lock(partition.sparse_cache_mutex);
      const auto [entry, inserted] = partition.sparse_cache.try_emplace(key, std::move(buffer));
      pointer = entry->second.as<uint64_t>();  // another thread may have been first: use its plane, ours returns to the pool
   }
   return std::make_unique<operators::IndexScan>(pointer, rows);
}

}  // namespace

namespace {

/// The nucleotide sequence a leaf names, or the database's default one; unknown names are the caller's mistake (400).
const std::string& nucleotideSequenceOf(const Database& database, const std::optional<std::string>& requested) {
   const std::string& name = requested.has_value() ? *requested : database.database_config.default_nucleotide_sequence;
   CHECK_SILO_QUERY(database.nuc_sequences.count(name) != 0, "Database does not contain the nucleotide sequence with name: '" + name + "'")
   return name;
}

/// "One of these symbols at the position": the Or of the per-symbol leaves, compiled without ambiguity.
template <typename Leaf, typename Symbols>
std::unique_ptr<Operator> anyOfSymbols(
   const Database& database, const DatabasePartition& partition, const std::string& sequence_name, uint32_t position, const Symbols& symbols
) {
   ExpressionVector leaves;
   for (const auto symbol : symbols) {
      leaves.push_back(std::make_unique<Leaf>(sequence_name, position, symbol));
   }
   return Or(std::move(leaves)).compile(database, partition, Expression::NONE);
}

/// "Not the reference symbol", the upper bound of "has a mutation" (has_mutation.cpp:51-56, has_aa_mutation.cpp:40-45).
template <typename Leaf, typename Symbol>
std::unique_ptr<Operator> notTheReference(
   const Database& database, const DatabasePartition& partition, const std::string& sequence_name, uint32_t position, Symbol reference
) {
   return Negation(std::make_unique<Leaf>(sequence_name, position, reference)).compile(database, partition, Expression::NONE);
}

}  // namespace

// ---- NucleotideSymbolEquals (nucleotide_symbol_equals.cpp:94-189) -----------------------------------
std::string NucleotideSymbolEquals::toString(const Database& /*database*/) const {
   const std::string prefix = nuc_sequence_name ? nuc_sequence_name.value() + ":" : "";
   const char symbol_char = value.has_value() ? Nucleotide::symbolToChar(*value) : '.';
   return prefix + std::to_string(position + 1) + std::to_string(symbol_char);
}

std::unique_ptr<Operator> NucleotideSymbolEquals::compile(
   const Database& database, const DatabasePartition& database_partition, AmbiguityMode mode
) const {
   const std::string& sequence_name = nucleotideSequenceOf(database, nuc_sequence_name);
   const auto& store = database_partition.nuc_sequences.at(sequence_name);
   const size_t genome_length = store.reference_sequence.size();
   CHECK_SILO_QUERY(
      position < genome_length,
      "NucleotideEquals position is out of bounds '" + std::to_string(position + 1) + "' > '" + std::to_string(genome_length) + "'"
   )
   const Nucleotide::Symbol wanted = value.has_value() ? *value : store.reference_sequence[position];  // "." asks for the reference symbol
   if (mode != UPPER_BOUND) {
      return symbolPlane<Nucleotide>(database, store, database_partition, position, wanted);
   }
   // the symbol or any ambiguity code that may stand for it (nucleotide_symbol_equals.cpp:137-149)
   return cachedUpperBoundPlane(database, database_partition, store.seqstore_id, position, static_cast<uint32_t>(wanted), [&]() {
      return anyOfSymbols<NucleotideSymbolEquals>(
         database, database_partition, sequence_name, position, AMBIGUITY_NUC_SYMBOLS.at(static_cast<uint32_t>(wanted))
      );
   });
}

// ---- AASymbolEquals (aa_symbol_equals.cpp:41-92; the ambiguity mode is ignored, :44) ----------------
std::string AASymbolEquals::toString(const Database& /*database*/) const {
   const char symbol_char = value.has_value() ? AminoAcid::symbolToChar(*value) : '.';
   return aa_sequence_name + ":" + std::to_string(position + 1) + std::to_string(symbol_char);
}

std::unique_ptr<Operator> AASymbolEquals::compile(
   const Database& database, const DatabasePartition& database_partition, AmbiguityMode /*mode*/
) const {
   const auto& store = database_partition.aa_sequences.at(aa_sequence_name);  // unknown gene: std::out_of_range -> 500, as in the reference
   const size_t gene_length = store.reference_sequence.size();
   CHECK_SILO_QUERY(
      position < gene_length,
      "AminoAcidEquals position is out of bounds '" + std::to_string(position + 1) + "' > '" + std::to_string(gene_length) + "'"
   )
   // The reference's rewrite for a deleted STOP symbol recurses forever; the dense
   // store simply returns the intended set.
   return symbolPlane<AminoAcid>(
      database, store, database_partition, position, value.has_value() ? *value : store.reference_sequence[position]
   );
}

// ---- SILO_COMPAT_REMOVE_QUIRK ------------------------------------------------------------------------
// HasMutation / HasAAMutation list "every symbol but the reference (and the missing symbol)" and OR the listed symbols.
// The reference takes a symbol off the list with std::remove WITHOUT the erase that has to follow it
// (has_mutation.cpp:58-65, has_aa_mutation.cpp:48-52): the kept entries are shifted to the front, the list keeps its
// length and its tail keeps what was there — so the last entry survives, duplicated or not.  Observable effect: where
// the reference symbol is the LAST entry (T among A C G T; STOP among the amino-acid symbols) it stays in the list, and
// HasNucleotideMutation at a reference-T position also matches the rows that carry T.  Parity is judged against the
// reference as it is, so this is what the engine does by default; Database::compat_remove_quirk = false (compile-time
// default SILO_COMPAT_REMOVE_QUIRK, run-time option "compat_remove_quirk") gives the list the code meant to build.
namespace {
template <typename Symbol>
void dropSymbol(std::vector<Symbol>& symbols, Symbol unwanted, bool as_the_reference_does) {
   const auto kept_end = std::remove(symbols.begin(), symbols.end(), unwanted);
   if (!as_the_reference_does) {
      symbols.erase(kept_end, symbols.end());
   }
}
}  // namespace

// ---- HasMutation (has_mutation.cpp:35-78) ----------------------------------------------------------
std::string HasMutation::toString(const Database& /*database*/) const {
   const std::string prefix = nuc_sequence_name ? nuc_sequence_name.value() + ":" : "";
   return prefix + std::to_string(position);
}

std::unique_ptr<Operator> HasMutation::compile(const Database& database, const DatabasePartition& database_partition, AmbiguityMode mode) const {
   const std::string& sequence_name = nucleotideSequenceOf(database, nuc_sequence_name);
   const Nucleotide::Symbol reference = database.nuc_sequences.at(sequence_name).reference_sequence.at(position);  // position 0 wrapped: 500
   if (mode == UPPER_BOUND) {
      return notTheReference<NucleotideSymbolEquals>(database, database_partition, sequence_name, position, reference);
   }
   std::vector<Nucleotide::Symbol> substitutions = {NS::A, NS::C, NS::G, NS::T};  // a gap is not a mutation here
   dropSymbol(substitutions, reference, database.compat_remove_quirk);  // has_mutation.cpp:58-65
   return anyOfSymbols<NucleotideSymbolEquals>(database, database_partition, sequence_name, position, substitutions);
}

// ---- HasAAMutation (has_aa_mutation.cpp:33-63) -----------------------------------------------------
std::string HasAAMutation::toString(const Database& /*database*/) const {
   return aa_sequence_name + ":" + std::to_string(position);
}

std::unique_ptr<Operator> HasAAMutation::compile(const Database& database, const DatabasePartition& database_partition, AmbiguityMode mode) const {
   const AminoAcid::Symbol reference = database.aa_sequences.at(aa_sequence_name).reference_sequence.at(position);
   if (mode == UPPER_BOUND) {
      return notTheReference<AASymbolEquals>(database, database_partition, aa_sequence_name, position, reference);
   }
   std::vector<AminoAcid::Symbol> substitutions(AminoAcid::SYMBOLS.begin(), AminoAcid::SYMBOLS.end());
   dropSymbol(substitutions, AminoAcid::Symbol::X, database.compat_remove_quirk);  // has_aa_mutation.cpp:48-52
   dropSymbol(substitutions, reference, database.compat_remove_quirk);
   return anyOfSymbols<AASymbolEquals>(database, database_partition, aa_sequence_name, position, substitutions);
}

// ---- PangoLineageFilter (pango_lineage_filter.cpp:37-59) ---------------------------------------------
std::string PangoLineageFilter::toString(const Database& /*database*/) const {
   return include_sublineages ? lineage + "*" : lineage;
}

std::unique_ptr<Operator> PangoLineageFilter::compile(
   const Database& /*database*/, const DatabasePartition& database_partition, AmbiguityMode /*mode*/
) const {
   const RowSpace rows = rowsOf(database_partition);
   const auto found = database_partition.columns.pango_lineage_columns.find(column);
   if (found == database_partition.columns.pango_lineage_columns.end()) {
      return std::make_unique<operators::Empty>(rows);
   }
   std::string lineage_all_upper = lineage;
   std::transform(lineage_all_upper.begin(), lineage_all_upper.end(), lineage_all_upper.begin(), ::toupper);
   const auto& pango_lineage_column = found->second;
   const auto bitmap = include_sublineages ? pango_lineage_column.filterIncludingSublineages(lineage_all_upper)
                                           : pango_lineage_column.filter(lineage_all_upper);
   if (bitmap == std::nullopt) {
      return std::make_unique<operators::Empty>(rows);
   }
   return std::make_unique<operators::IndexScan>(bitmap.value(), rows);
}

// ---- metadata predicates ------------------------------------------------------
namespace {

operators::Predicate predicateOf(const storage::column::MetadataColumnPartition& column, int comparator) {
   operators::Predicate predicate{};
   predicate.column = &column;
   predicate.comparator = comparator;
   return predicate;
}
operators::Predicate intPredicate(const storage::column::MetadataColumnPartition& column, int comparator, int32_t value) {
   operators::Predicate predicate = predicateOf(column, comparator);
   predicate.value.as_int = value;
   return predicate;
}
operators::Predicate wordPredicate(const storage::column::MetadataColumnPartition& column, int comparator, uint32_t value) {
   operators::Predicate predicate = predicateOf(column, comparator);
   predicate.value.as_word = value;
   return predicate;
}
operators::Predicate doublePredicate(const storage::column::MetadataColumnPartition& column, int comparator, double value) {
   operators::Predicate predicate = predicateOf(column, comparator);
   predicate.value.as_double = value;
   return predicate;
}
std::unique_ptr<Operator> selectionOf(std::vector<operators::Predicate> predicates, RowSpace rows) {
   return std::make_unique<operators::Selection>(std::move(predicates), rows);
}

}  // namespace

std::string StringEquals::toString(const Database& /*database*/) const {
   return column + " = '" + value + "'";
}
std::unique_ptr<Operator> StringEquals::compile(
   const Database& /*database*/, const DatabasePartition& database_partition, AmbiguityMode /*mode*/
) const {  // string_equals.cpp:37-68
   const RowSpace rows = rowsOf(database_partition);
   const auto* indexed = database_partition.columns.find(column, config::ColumnType::INDEXED_STRING);
   const auto* plain = database_partition.columns.find(column, config::ColumnType::STRING);
   const auto* string_column = indexed != nullptr ? indexed : plain;
   if (string_column == nullptr) {
      return std::make_unique<operators::Empty>(rows);
   }
   // Both kinds are dictionary encoded here: the filter is "dictionary id == id of the value", one compare pass
   // over 4 bytes per row (the reference keeps a roaring bitmap per value for indexed columns and probes the
   // embedded strings row by row for plain ones).  A value that is not in the dictionary matches no row.
   const auto value_id = string_column->lookupId(value);
   if (!value_id.has_value()) {
      return std::make_unique<operators::Empty>(rows);
   }
   if (indexed != nullptr) {
      // an indexed column answers with a stored bitmap (IndexScan, string_equals.cpp:45-55): the bitset of a value is
      // built by one compare pass on first use and kept
      const auto key = std::make_pair(indexed, *value_id);
      const size_t row_bytes = static_cast<size_t>(database_partition.rowWords()) * sizeof(uint64_t);
      bool room = false;
      {
         const std::shared_lock<std::shared_mutex> lock(database_partition.sparse_cache_mutex);
         const auto found = database_partition.indexed_value_cache.find(key);
         if (found != database_partition.indexed_value_cache.end()) {
            return std::make_unique<operators::IndexScan>(found->second.as<uint64_t>(), rows);
         }
         room = (database_partition.indexed_value_cache.size() + 1) * row_bytes <= DatabasePartition::INDEXED_VALUE_CACHE_BYTES;
      }
      if (room) {
         DeviceBuffer buffer = database_partition.pool.acquire(row_bytes);
         const uint32_t id = *value_id;
         checkGpu(
            silo_gpu_bitset_from_compare(
               database_partition.store, buffer.as<uint64_t>(), indexed->deviceValues(), SILO_GPU_VALUE_U32, SILO_GPU_CMP_EQUALS, &id, queryStream()
            ),
            "silo_gpu_bitset_from_compare"
         );
         checkGpu(silo_gpu_stream_synchronize(queryStream()), "silo_gpu_stream_synchronize");  // other streams may read it at once
         const std::unique_lock<std::shared_mutex> lock(database_partition.sparse_cache_mutex);
         const auto [entry, inserted] = database_partition.indexed_value_cache.try_emplace(key, std::move(buffer));
         return std::make_unique<operators::IndexScan>(entry->second.as<uint64_t>(), rows);
      }
   }
   return selectionOf({wordPredicate(*string_column, SILO_GPU_CMP_EQUALS, *value_id)}, rows);
}

std::string IntEquals::toString(const Database& /*database*/) const {
   return column + " = '" + std::to_string(value) + "'";
}
std::unique_ptr<Operator> IntEquals::compile(
   const Database& /*database*/, const DatabasePartition& database_partition, AmbiguityMode /*mode*/
) const {  // int_equals.cpp:30-47
   const RowSpace rows = rowsOf(database_partition);
   const auto* int_column = database_partition.columns.find(column, config::ColumnType::INT);
   if (int_column == nullptr) {
      return std::make_unique<operators::Empty>(rows);
   }
   return selectionOf({intPredicate(*int_column, SILO_GPU_CMP_EQUALS, value)}, rows);
}

std::string IntBetween::toString(const Database& /*database*/) const {
   return "[IntBetween " + (from.has_value() ? std::to_string(*from) : "unbounded") + " - " + (to.has_value() ? std::to_string(*to) : "unbounded") + "]";
}
std::unique_ptr<Operator> IntBetween::compile(
   const Database& /*database*/, const DatabasePartition& database_partition, AmbiguityMode /*mode*/
) const {  // int_between.cpp:37-60
   const auto* int_column = database_partition.columns.find(column, config::ColumnType::INT);
   if (int_column == nullptr) {
      throw std::out_of_range("map::at");  // int_columns.at(column): an unknown column is a 500 in the reference
   }
   std::vector<operators::Predicate> predicates;
   predicates.push_back(intPredicate(*int_column, SILO_GPU_CMP_HIGHER_OR_EQUALS, from.value_or(INT32_MIN + 1)));
   if (to.has_value()) {
      predicates.push_back(intPredicate(*int_column, SILO_GPU_CMP_LESS_OR_EQUALS, to.value()));
   }
   return selectionOf(std::move(predicates), rowsOf(database_partition));
}

std::string FloatEquals::toString(const Database& /*database*/) const {
   return column + " = '" + std::to_string(value) + "'";
}
std::unique_ptr<Operator> FloatEquals::compile(
   const Database& /*database*/, const DatabasePartition& database_partition, AmbiguityMode /*mode*/
) const {  // float_equals.cpp:33-50
   const RowSpace rows = rowsOf(database_partition);
   const auto* float_column = database_partition.columns.find(column, config::ColumnType::FLOAT);
   if (float_column == nullptr) {
      return std::make_unique<operators::Empty>(rows);
   }
   return selectionOf({doublePredicate(*float_column, SILO_GPU_CMP_EQUALS, value)}, rows);
}

std::string FloatBetween::toString(const Database& /*database*/) const {
   return "[FloatBetween " + (from.has_value() ? std::to_string(*from) : "unbounded") + " - " + (to.has_value() ? std::to_string(*to) : "unbounded") + "]";
}
std::unique_ptr<Operator> FloatBetween::compile(
   const Database& /*database*/, const DatabasePartition& database_partition, AmbiguityMode /*mode*/
) const {  // float_between.cpp:37-69: [from, to) — the upper bound is exclusive
   const auto* float_column = database_partition.columns.find(column, config::ColumnType::FLOAT);
   CHECK_SILO_QUERY(float_column != nullptr, "The database does not contain the float column '" + column + "'")
   std::vector<operators::Predicate> predicates;
   if (from.has_value()) {
      predicates.push_back(doublePredicate(*float_column, SILO_GPU_CMP_HIGHER_OR_EQUALS, from.value()));
   }
   if (to.has_value()) {
      predicates.push_back(doublePredicate(*float_column, SILO_GPU_CMP_LESS, to.value()));
   }
   if (predicates.empty()) {
      predicates.push_back(doublePredicate(*float_column, SILO_GPU_CMP_NOT_EQUALS, std::nan("")));  // true for every row
   }
   return selectionOf(std::move(predicates), rowsOf(database_partition));
}

std::string DateBetween::toString(const Database& /*database*/) const {
   return "[Date-between " + (date_from.has_value() ? common::dateToString(*date_from).value_or("") : "unbounded") + " and " +
          (date_to.has_value() ? common::dateToString(*date_to).value_or("") : "unbounded") + "]";
}
std::unique_ptr<Operator> DateBetween::compile(
   const Database& /*database*/, const DatabasePartition& database_partition, AmbiguityMode /*mode*/
) const {  // date_between.cpp:49-101
   const auto* date_column = database_partition.columns.find(column, config::ColumnType::DATE);
   if (date_column == nullptr) {
      throw std::out_of_range("map::at");  // date_columns.at(column)
   }
   std::vector<operators::Predicate> predicates;
   if (!date_column->is_sorted) {
      // from <= d < to: on an unsorted column the upper bound is EXCLUSIVE (:58-73)
      predicates.push_back(wordPredicate(*date_column, SILO_GPU_CMP_HIGHER_OR_EQUALS, date_from.value_or(common::Date{1})));
      predicates.push_back(wordPredicate(*date_column, SILO_GPU_CMP_LESS, date_to.value_or(common::Date{UINT32_MAX})));
   } else {
      // the dateToSortBy column: the reference binary-searches its physically sorted rows, lower_bound(from or 1) to
      // upper_bound(to) — from <= d <= to, the upper bound INCLUSIVE, NULL (0) excluded (:83-101).  Rows keep their
      // input order here, so the same set comes from two compare passes instead of id ranges.
      predicates.push_back(wordPredicate(*date_column, SILO_GPU_CMP_HIGHER_OR_EQUALS, date_from.value_or(common::Date{1})));
      if (date_to.has_value()) {
         predicates.push_back(wordPredicate(*date_column, SILO_GPU_CMP_LESS_OR_EQUALS, date_to.value()));
      }
   }
   return selectionOf(std::move(predicates), rowsOf(database_partition));
}

// ---- InsertionContains (insertion_contains.cpp) ------------------------------------------------------
template <typename SymbolType>
std::string InsertionContains<SymbolType>::toString(const Database& /*database*/) const {
   const std::string symbol_name = std::string(SymbolType::SYMBOL_NAME);
   const std::string sequence_string =
      sequence_name.has_value() ? "The sequence '" + sequence_name.value() + "'" : "The default " + symbol_name + " sequence ";
   return sequence_string + " has insertion '" + value + "'";
}

template <typename SymbolType>
std::unique_ptr<Operator> InsertionContains<SymbolType>::compile(
   const Database& database, const DatabasePartition& database_partition, AmbiguityMode /*mode*/
) const {  // insertion_contains.cpp:65-131
   const RowSpace rows = rowsOf(database_partition);
   const auto& insertion_columns = database_partition.columns.getInsertionColumns<SymbolType>();
   for (const std::string& column_name : column_names) {
      CHECK_SILO_QUERY(insertion_columns.count(column_name) != 0, "The insertion column '" + column_name + "' does not exist.")
   }
   if (insertion_columns.empty()) {
      return std::make_unique<operators::Empty>(rows);
   }
   std::string validated_sequence_name;
   if (sequence_name.has_value()) {
      validated_sequence_name = sequence_name.value();
   } else {
      // only nucleotide sequences have a default (database.cpp:73-80)
      CHECK_SILO_QUERY(
         (std::is_same_v<SymbolType, Nucleotide>), "The database has no default " + std::string(SymbolType::SYMBOL_NAME_LOWER_CASE) + " sequence name"
      )
      validated_sequence_name = database.database_config.default_nucleotide_sequence;
   }
   // InsertionIndex::search (insertion_index.cpp:271-281): the reference pre-selects candidates through a 3-mer index
   // and then runs regex_search on them; the answer is regex_search over the distinct insertions at the position.
   // That part stays on the host (a handful of short strings); the rows are gathered on the device.
   const std::regex search_pattern(value);
   OperatorVector column_operators;
   for (const auto& [column_name, insertion_column] : insertion_columns) {
      if (!column_names.empty() && std::find(column_names.begin(), column_names.end(), column_name) == column_names.end()) {
         continue;
      }
      const auto found = insertion_column.getInsertionIndexes().find(validated_sequence_name);
      if (found == insertion_column.getInsertionIndexes().end()) {
         continue;
      }
      const auto& index = found->second;
      std::vector<uint8_t> membership(index.insertions.size(), 0);
      const auto at_position = index.ids_at_position.find(position);
      if (at_position != index.ids_at_position.end()) {
         for (const uint32_t id : at_position->second) {
            membership[id] = std::regex_search(index.insertions[id], search_pattern) ? 1 : 0;
         }
      }
      column_operators.emplace_back(std::make_unique<operators::BitmapProducer>(&index, std::move(membership), rows));
   }
   if (column_operators.empty()) {
      return std::make_unique<operators::Empty>(rows);
   }
   if (column_operators.size() == 1) {
      return std::move(column_operators.at(0));
   }
   return std::make_unique<operators::Union>(std::move(column_operators), rows);
}

template struct InsertionContains<Nucleotide>;
template struct InsertionContains<AminoAcid>;

namespace {

template <typename SymbolType>
std::unique_ptr<Expression> parseInsertionContains(const json::Value& json) {  // insertion_contains.cpp:155-214
   CHECK_SILO_QUERY(
      !json.contains("column") || (json["column"].is_string() || json["column"].is_array()),
      "The InsertionsContains filter can have the field column of type string or an array of strings, but no other type"
   )
   std::vector<std::string> column_names;
   if (json.contains("column") && json["column"].is_array()) {
      for (const auto& child : json["column"].items()) {
         CHECK_SILO_QUERY(
            child.is_string(), "The field column of the InsertionsContains filter must have type string or an array, if present. Found:" + child.dump()
         )
         column_names.emplace_back(child.as_string());
      }
   } else if (json.contains("column") && json["column"].is_string()) {
      column_names.emplace_back(json["column"].as_string());
   }
   CHECK_SILO_QUERY(json.contains("position"), "The field 'position' is required in an InsertionContains expression")
   CHECK_SILO_QUERY(
      json["position"].is_number_unsigned() && (json["position"].as_uint32() > 0),
      "The field 'position' in an InsertionContains expression needs to be a positive number (> 0)"
   )
   CHECK_SILO_QUERY(
      !json.contains("sequenceName") || json["sequenceName"].is_string(),
      "The optional field 'sequenceName' in an InsertionContains expression needs to be a string"
   )
   CHECK_SILO_QUERY(json.contains("value"), "The field 'value' is required in an InsertionContains expression")
   CHECK_SILO_QUERY(json["value"].is_string(), "The field 'value' in an InsertionContains expression needs to be a string")
   std::optional<std::string> sequence_name;
   if (json.contains("sequenceName")) {
      sequence_name = json["sequenceName"].as_string();
   }
   const uint32_t position = json["position"].as_uint32();
   const std::string& value = json["value"].as_string();
   CHECK_SILO_QUERY(!value.empty(), "The field 'value' in an InsertionContains expression must not be an empty string")
   // ^([symbols]|\.\*)*$ (:133-149)
   std::string valid_pattern = "^([";
   for (const auto symbol : SymbolType::SYMBOLS) {
      valid_pattern += SymbolType::symbolToChar(symbol);
   }
   valid_pattern += "]|\\.\\*)*$";
   CHECK_SILO_QUERY(
      std::regex_search(value, std::regex(valid_pattern)),
      "The field 'value' in the InsertionContains expression does not contain a valid regex pattern: \"" + value +
         "\". It must only consist of " + std::string(SymbolType::SYMBOL_NAME_LOWER_CASE) + " symbols and the regex symbol '.*'."
   )
   return std::make_unique<InsertionContains<SymbolType>>(std::move(column_names), sequence_name, position, value);
}

}  // namespace

// ---- JSON -> Expression (the from_json functions) ------------------------------------------------------
namespace {

ExpressionVector parseChildren(const json::Value& json) {
   ExpressionVector children;
   for (const auto& child : json.at("children").items()) {
      children.push_back(parseExpression(child));
   }
   return children;
}

}  // namespace

std::unique_ptr<Expression> parseExpression(const json::Value& json) {  // expression.cpp:49-102
   CHECK_SILO_QUERY(json.contains("type"), "The field 'type' is required in any filter expression")
   CHECK_SILO_QUERY(
      json["type"].is_string(), "The field 'type' in all filter expressions needs to be a string, but is: " + json["type"].dump()
   )
   const std::string& expression_type = json["type"].as_string();
   if (expression_type == "True") {
      return std::make_unique<True>();
   }
   if (expression_type == "False") {
      return std::make_unique<False>();
   }
   if (expression_type == "And") {  // and.cpp:230-239
      CHECK_SILO_QUERY(json.contains("children"), "The field 'children' is required in an And expression")
      CHECK_SILO_QUERY(json["children"].is_array(), "The field 'children' in an And expression needs to be an array")
      return std::make_unique<And>(parseChildren(json));
   }
   if (expression_type == "Or") {  // or.cpp:97-106
      CHECK_SILO_QUERY(json.contains("children"), "The field 'children' is required in an Or expression")
      CHECK_SILO_QUERY(json["children"].is_array(), "The field 'children' in an Or expression needs to be an array")
      return std::make_unique<Or>(parseChildren(json));
   }
   if (expression_type == "N-Of") {  // nof.cpp:283-316
      CHECK_SILO_QUERY(json.contains("children"), "The field 'children' is required in an N-Of expression")
      CHECK_SILO_QUERY(json["children"].is_array(), "The field 'children' in an N-Of expression needs to be an array")
      CHECK_SILO_QUERY(json.contains("numberOfMatchers"), "The field 'numberOfMatchers' is required in an N-Of expression")
      CHECK_SILO_QUERY(
         json["numberOfMatchers"].is_number_unsigned(), "The field 'numberOfMatchers' in an N-Of expression needs to be an unsigned integer"
      )
      CHECK_SILO_QUERY(json.contains("matchExactly"), "The field 'matchExactly' is required in an N-Of expression")
      CHECK_SILO_QUERY(json["matchExactly"].is_boolean(), "The field 'matchExactly' in an N-Of expression needs to be a boolean")
      const uint32_t number_of_matchers = json["numberOfMatchers"].as_uint32();
      const bool match_exactly = json["matchExactly"].as_bool();
      return std::make_unique<NOf>(parseChildren(json), static_cast<int>(number_of_matchers), match_exactly);
   }
   if (expression_type == "Not") {
      CHECK_SILO_QUERY(json.contains("child"), "The field 'child' is required in a Not expression")
      return std::make_unique<Negation>(parseExpression(json["child"]));
   }
   if (expression_type == "Maybe") {
      CHECK_SILO_QUERY(json.contains("child"), "The field 'child' is required in a Maybe expression")
      return std::make_unique<Maybe>(parseExpression(json["child"]));
   }
   if (expression_type == "Exact") {
      CHECK_SILO_QUERY(json.contains("child"), "The field 'child' is required in a Exact expression")
      return std::make_unique<Exact>(parseExpression(json["child"]));
   }
   if (expression_type == "NucleotideEquals") {  // nucleotide_symbol_equals.cpp:192-227
      CHECK_SILO_QUERY(json.is_object() && json.contains("position"), "The field 'position' is required in a NucleotideEquals expression")
      CHECK_SILO_QUERY(
         json["position"].is_number_unsigned() && json["position"].as_uint32() > 0,
         "The field 'position' in a NucleotideEquals expression needs to be an unsigned integer greater than 0"
      )
      CHECK_SILO_QUERY(json.contains("symbol"), "The field 'symbol' is required in a NucleotideEquals expression")
      CHECK_SILO_QUERY(json["symbol"].is_string(), "The field 'symbol' in a NucleotideEquals expression needs to be a string")
      std::optional<std::string> nuc_sequence_name;
      if (json.contains("sequenceName")) {
         nuc_sequence_name = json["sequenceName"].as_string();
      }
      const uint32_t position = json["position"].as_uint32() - 1;
      const std::string& nucleotide_symbol = json["symbol"].as_string();
      CHECK_SILO_QUERY(nucleotide_symbol.size() == 1, "The string field 'symbol' must be exactly one character long")
      const std::optional<Nucleotide::Symbol> nuc_value = Nucleotide::charToSymbol(nucleotide_symbol.at(0));
      CHECK_SILO_QUERY(
         nuc_value.has_value() || nucleotide_symbol.at(0) == '.',
         "The string field 'symbol' must be either a valid nucleotide symbol or the '.' symbol."
      )
      return std::make_unique<NucleotideSymbolEquals>(nuc_sequence_name, position, nuc_value);
   }
   if (expression_type == "AminoAcidEquals") {  // aa_symbol_equals.cpp:95-125
      CHECK_SILO_QUERY(
         json.contains("sequenceName") && json["sequenceName"].is_string(), "AminoAcidEquals expression requires the string field sequenceName"
      )
      CHECK_SILO_QUERY(json.is_object() && json.contains("position"), "The field 'position' is required in a AminoAcidEquals expression")
      CHECK_SILO_QUERY(
         json["position"].is_number_unsigned() && json["position"].as_uint32() > 0,
         "The field 'position' in a AminoAcidEquals expression needs to be an unsigned integer greater than 0"
      )
      CHECK_SILO_QUERY(
         json.contains("symbol") && json["symbol"].is_string(), "The string field 'symbol' is required in a AminoAcidEquals expression"
      )
      const std::string aa_sequence_name = json["sequenceName"].as_string();
      const uint32_t position = json["position"].as_uint32() - 1;
      const std::string aa_char = json["symbol"].as_string();
      CHECK_SILO_QUERY(aa_char.size() == 1, "The string field 'symbol' must be exactly one character long")
      const std::optional<AminoAcid::Symbol> aa_value = AminoAcid::charToSymbol(aa_char.at(0));
      CHECK_SILO_QUERY(
         aa_value.has_value() || aa_char.at(0) == '.', "The string field 'symbol' must be either a valid amino acid or the '.' symbol."
      )
      return std::make_unique<AASymbolEquals>(aa_sequence_name, position, aa_value);
   }
   if (expression_type == "HasNucleotideMutation") {  // has_mutation.cpp:81-96
      CHECK_SILO_QUERY(json.contains("position"), "The field 'position' is required in a HasNucleotideMutation expression")
      CHECK_SILO_QUERY(
         json["position"].is_number_unsigned(), "The field 'position' in a HasNucleotideMutation expression needs to be an unsigned integer"
      )
      std::optional<std::string> nuc_sequence_name;
      if (json.contains("sequenceName")) {
         nuc_sequence_name = json["sequenceName"].as_string();
      }
      const uint32_t position = json["position"].as_uint32() - 1;  // position 0 wraps, then .at() throws -> 500
      return std::make_unique<HasMutation>(nuc_sequence_name, position);
   }
   if (expression_type == "HasAminoAcidMutation") {  // has_aa_mutation.cpp:66-84
      CHECK_SILO_QUERY(json.contains("position"), "The field 'position' is required in a HasAminoAcidMutation expression")
      CHECK_SILO_QUERY(
         json["position"].is_number_unsigned(), "The field 'position' in a HasAminoAcidMutation expression needs to be an unsigned integer"
      )
      CHECK_SILO_QUERY(
         json.contains("sequenceName") && json["sequenceName"].is_string(),
         "HasAminoAcidMutation expression requires the string field sequenceName"
      )
      const std::string aa_sequence_name = json["sequenceName"].as_string();
      const uint32_t position = json["position"].as_uint32() - 1;
      return std::make_unique<HasAAMutation>(aa_sequence_name, position);
   }
   if (expression_type == "PangoLineage") {  // pango_lineage_filter.cpp:62-92
      CHECK_SILO_QUERY(json.contains("column"), "The field 'column' is required in a PangoLineage expression")
      CHECK_SILO_QUERY(json["column"].is_string(), "The field 'column' in a PangoLineage expression needs to be a string")
      CHECK_SILO_QUERY(json.contains("value"), "The field 'value' is required in a PangoLineage expression")
      CHECK_SILO_QUERY(json["value"].is_string(), "The field 'value' in a PangoLineage expression needs to be a string")
      CHECK_SILO_QUERY(json.contains("includeSublineages"), "The field 'includeSublineages' is required in a PangoLineage expression")
      CHECK_SILO_QUERY(
         json["includeSublineages"].is_boolean(), "The field 'includeSublineages' in a PangoLineage expression needs to be a boolean"
      )
      return std::make_unique<PangoLineageFilter>(json["column"].as_string(), json["value"].as_string(), json["includeSublineages"].as_bool());
   }
   if (expression_type == "StringEquals") {  // string_equals.cpp:70-85
      CHECK_SILO_QUERY(json.contains("column"), "The field 'column' is required in an StringEquals expression")
      CHECK_SILO_QUERY(json["column"].is_string(), "The field 'column' in an StringEquals expression needs to be a string")
      CHECK_SILO_QUERY(json.contains("value"), "The field 'value' is required in an StringEquals expression")
      CHECK_SILO_QUERY(
         json["value"].is_string() || json["value"].is_null(), "The field 'value' in an StringEquals expression needs to be a string or null"
      )
      return std::make_unique<StringEquals>(json["column"].as_string(), json["value"].is_null() ? "" : json["value"].as_string());
   }
   if (expression_type == "IntEquals") {  // int_equals.cpp:50-67
      CHECK_SILO_QUERY(json.contains("column"), "The field 'column' is required in an IntEquals expression")
      CHECK_SILO_QUERY(json["column"].is_string(), "The field 'column' in an IntEquals expression must be a string")
      CHECK_SILO_QUERY(json.contains("value"), "The field 'value' is required in an IntEquals expression")
      CHECK_SILO_QUERY(
         json["value"].is_number_integer() || json["value"].is_null(), "The field 'value' in an IntEquals expression must be an integer or null"
      )
      return std::make_unique<IntEquals>(
         json["column"].as_string(), json["value"].is_null() ? INT32_MIN : static_cast<int32_t>(json["value"].as_int64())
      );
   }
   if (expression_type == "IntBetween") {  // int_between.cpp:63-88
      CHECK_SILO_QUERY(json.contains("column"), "The field 'column' is required in a IntBetween expression")
      CHECK_SILO_QUERY(json["column"].is_string(), "The field 'column' in a IntBetween expression must be a string")
      CHECK_SILO_QUERY(json.contains("from"), "The field 'from' is required in IntBetween expression")
      CHECK_SILO_QUERY(
         json["from"].is_null() || json["from"].is_number_integer(), "The field 'from' in a IntBetween expression must be an int or null"
      )
      CHECK_SILO_QUERY(json.contains("to"), "The field 'to' is required in a IntBetween expression")
      CHECK_SILO_QUERY(json["to"].is_null() || json["to"].is_number_integer(), "The field 'to' in a IntBetween expression must be an int or null")
      std::optional<int32_t> value_from;
      if (json["from"].is_number_integer()) {
         value_from = static_cast<int32_t>(json["from"].as_int64());
      }
      std::optional<int32_t> value_to;
      if (json["to"].is_number_integer()) {
         value_to = static_cast<int32_t>(json["to"].as_int64());
      }
      return std::make_unique<IntBetween>(json["column"].as_string(), value_from, value_to);
   }
   if (expression_type == "FloatEquals") {  // float_equals.cpp:53-70
      CHECK_SILO_QUERY(json.contains("column"), "The field 'column' is required in an FloatEquals expression")
      CHECK_SILO_QUERY(json["column"].is_string(), "The field 'column' in an FloatEquals expression must be a string")
      CHECK_SILO_QUERY(json.contains("value"), "The field 'value' is required in an FloatEquals expression")
      CHECK_SILO_QUERY(json["value"].is_number_float() || json["value"].is_null(), "The field 'value' in an FloatEquals expression must be a float")
      return std::make_unique<FloatEquals>(json["column"].as_string(), json["value"].is_null() ? std::nan("") : json["value"].as_double());
   }
   if (expression_type == "FloatBetween") {  // float_between.cpp:72-99
      CHECK_SILO_QUERY(json.contains("column"), "The field 'column' is required in a FloatBetween expression")
      CHECK_SILO_QUERY(json["column"].is_string(), "The field 'column' in a FloatBetween expression must be a string")
      CHECK_SILO_QUERY(json.contains("from"), "The field 'from' is required in FloatBetween expression")
      CHECK_SILO_QUERY(
         json["from"].is_null() || json["from"].is_number_float(), "The field 'from' in a FloatBetween expression must be a float or null"
      )
      CHECK_SILO_QUERY(json.contains("to"), "The field 'to' is required in a FloatBetween expression")
      CHECK_SILO_QUERY(json["to"].is_null() || json["to"].is_number_float(), "The field 'to' in a FloatBetween expression must be a float or null")
      std::optional<double> value_from;
      if (json["from"].is_number_float()) {
         value_from = json["from"].as_double();
      }
      std::optional<double> value_to;
      if (json["to"].is_number_float()) {
         value_to = json["to"].as_double();
      }
      return std::make_unique<FloatBetween>(json["column"].as_string(), value_from, value_to);
   }
   if (expression_type == "DateBetween") {  // date_between.cpp:103-130
      CHECK_SILO_QUERY(json.contains("column"), "The field 'column' is required in a DateBetween expression")
      CHECK_SILO_QUERY(json["column"].is_string(), "The field 'column' in a DateBetween expression needs to be a string")
      CHECK_SILO_QUERY(json.contains("from"), "The field 'from' is required in DateBetween expression")
      // nlohmann's empty() is false for every string, so "non-empty" never rejects anything (:112-115)
      CHECK_SILO_QUERY(json["from"].is_null() || json["from"].is_string(), "The field 'from' in a DateBetween expression needs to be a string or null")
      CHECK_SILO_QUERY(json.contains("to"), "The field 'to' is required in a DateBetween expression")
      CHECK_SILO_QUERY(
         json["to"].is_null() || json["to"].is_string(), "The field 'to' in a DateBetween expression needs to be a non-empty string or null"
      )
      std::optional<common::Date> date_from;
      if (json["from"].is_string()) {
         date_from = common::stringToDate(json["from"].as_string());
      }
      std::optional<common::Date> date_to;
      if (json["to"].is_string()) {
         date_to = common::stringToDate(json["to"].as_string());
      }
      return std::make_unique<DateBetween>(json["column"].as_string(), date_from, date_to);
   }
   if (expression_type == "InsertionContains") {
      return parseInsertionContains<Nucleotide>(json);
   }
   if (expression_type == "AminoAcidInsertionContains") {
      return parseInsertionContains<AminoAcid>(json);
   }
   throw QueryParseException("Unknown object filter type '" + expression_type + "'");
}

}  // namespace silo::query_engine::filter_expressions
