// filter_expressions.cpp — logical filter tree: JSON -> Expression -> compile() -> operators.
// Mirrors src/silo/query_engine/filter_expressions/*.cpp of the reference (cited per function),
// including the algebraic rewrites and the observable quirks listed in SURVEY.md §8(a).
#include <algorithm>

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

std::string join(const std::vector<std::string>& parts, const std::string& separator) {
   std::string out;
   for (size_t i = 0; i < parts.size(); ++i) {
      if (i != 0) {
         out += separator;
      }
      out += parts[i];
   }
   return out;
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

// ---- And (and.cpp:101-227; Selection predicates are outside this path) -----------------------------
std::string And::toString(const Database& database) const {
   std::vector<std::string> child_strings;
   for (const auto& child : children) {
      child_strings.push_back(child->toString(database));
   }
   return "And(" + join(child_strings, " & ") + ")";
}

std::tuple<OperatorVector, OperatorVector, std::vector<operators::Predicate>> And::compileChildren(
   const Database& database, const DatabasePartition& database_partition, AmbiguityMode mode
) const {
   OperatorVector all_child_operators;
   for (const auto& expression : children) {
      all_child_operators.push_back(expression->compile(database, database_partition, mode));
   }
   OperatorVector non_negated_child_operators;
   OperatorVector negated_child_operators;
   std::vector<operators::Predicate> predicates;
   // by index: the child of a Selection is appended to the list while it is walked (and.cpp:138-151)
   for (size_t index = 0; index < all_child_operators.size(); ++index) {
      auto& child = all_child_operators[index];
      if (child->type() == operators::FULL) {
         continue;
      }
      if (child->type() == operators::EMPTY) {
         OperatorVector empty;
         empty.emplace_back(std::make_unique<operators::Empty>(rowsOf(database_partition)));
         return {std::move(empty), OperatorVector(), std::vector<operators::Predicate>{}};
      }
      if (child->type() == operators::INTERSECTION) {
         auto* intersection_child = dynamic_cast<operators::Intersection*>(child.get());
         for (auto& grandchild : intersection_child->children) {
            non_negated_child_operators.push_back(std::move(grandchild));
         }
         for (auto& grandchild : intersection_child->negated_children) {
            negated_child_operators.push_back(std::move(grandchild));
         }
      } else if (child->type() == operators::COMPLEMENT) {
         negated_child_operators.emplace_back(child->negate());
      } else if (child->type() == operators::SELECTION) {
         auto* selection_child = dynamic_cast<operators::Selection*>(child.get());
         predicates.insert(predicates.end(), selection_child->predicates.begin(), selection_child->predicates.end());
         if (selection_child->child != nullptr) {
            std::unique_ptr<Operator> grandchild = std::move(selection_child->child);
            all_child_operators.push_back(std::move(grandchild));  // may reallocate: `child` is not used afterwards
         }
      } else {
         non_negated_child_operators.push_back(std::move(child));
      }
   }
   return {std::move(non_negated_child_operators), std::move(negated_child_operators), std::move(predicates)};
}

std::unique_ptr<Operator> And::compile(const Database& database, const DatabasePartition& database_partition, AmbiguityMode mode) const {
   auto [non_negated_child_operators, negated_child_operators, predicates] = compileChildren(database, database_partition, mode);
   const RowSpace rows = rowsOf(database_partition);
   if (non_negated_child_operators.empty() && negated_child_operators.empty()) {
      if (predicates.empty()) {
         return std::make_unique<operators::Full>(rows);
      }
      return std::make_unique<operators::Selection>(std::move(predicates), rows);
   }
   std::unique_ptr<Operator> index_arithmetic_operator;
   if (non_negated_child_operators.size() == 1 && negated_child_operators.empty()) {
      index_arithmetic_operator = std::move(non_negated_child_operators[0]);
   } else if (negated_child_operators.size() == 1 && non_negated_child_operators.empty()) {
      index_arithmetic_operator = std::make_unique<operators::Complement>(std::move(negated_child_operators[0]), rows);
   } else if (non_negated_child_operators.empty()) {
      auto union_ret = std::make_unique<operators::Union>(std::move(negated_child_operators), rows);
      index_arithmetic_operator = std::make_unique<operators::Complement>(std::move(union_ret), rows);
   } else {
      index_arithmetic_operator =
         std::make_unique<operators::Intersection>(std::move(non_negated_child_operators), std::move(negated_child_operators), rows);
   }
   if (predicates.empty()) {
      return index_arithmetic_operator;
   }
   return std::make_unique<operators::Selection>(std::move(index_arithmetic_operator), std::move(predicates), rows);
}

// ---- Or (or.cpp:41-94) ---------------------------------------------------------------------------
std::string Or::toString(const Database& database) const {
   std::vector<std::string> child_strings;
   for (const auto& child : children) {
      child_strings.push_back(child->toString(database));
   }
   return "Or(" + join(child_strings, " | ") + ")";
}

std::unique_ptr<Operator> Or::compile(const Database& database, const DatabasePartition& database_partition, AmbiguityMode mode) const {
   const RowSpace rows = rowsOf(database_partition);
   OperatorVector all_child_operators;
   for (const auto& expression : children) {
      all_child_operators.push_back(expression->compile(database, database_partition, mode));
   }
   OperatorVector filtered_child_operators;
   for (auto& child : all_child_operators) {
      if (child->type() == operators::EMPTY) {
         continue;
      }
      if (child->type() == operators::FULL) {
         return std::make_unique<operators::Full>(rows);
      }
      if (child->type() == operators::UNION) {
         auto* or_child = dynamic_cast<operators::Union*>(child.get());
         for (auto& grandchild : or_child->children) {
            filtered_child_operators.push_back(std::move(grandchild));
         }
      } else {
         filtered_child_operators.push_back(std::move(child));
      }
   }
   if (filtered_child_operators.empty()) {
      return std::make_unique<operators::Empty>(rows);
   }
   if (filtered_child_operators.size() == 1) {
      return std::move(filtered_child_operators[0]);
   }
   if (std::any_of(filtered_child_operators.begin(), filtered_child_operators.end(), [](const auto& child) {
          return child->type() == operators::COMPLEMENT;
       })) {
      return operators::Complement::fromDeMorgan(std::move(filtered_child_operators), rows);
   }
   return std::make_unique<operators::Union>(std::move(filtered_child_operators), rows);
}

// ---- Negation / Maybe / Exact ----------------------------------------------------------------------
std::string Negation::toString(const Database& database) const {
   return "!(" + child->toString(database) + ")";
}
std::unique_ptr<Operator> Negation::compile(const Database& database, const DatabasePartition& database_partition, AmbiguityMode mode) const {
   auto child_operator = child->compile(database, database_partition, invertMode(mode));  // negation.cpp:27-34
   return child_operator->negate();
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

// ---- N-Of (nof.cpp) ------------------------------------------------------------------------------
namespace {

std::unique_ptr<Operator> handleTrivialCases(  // nof.cpp:35-88
   const int updated_number_of_matchers, OperatorVector& non_negated_child_operators, OperatorVector& negated_child_operators,
   bool match_exactly, RowSpace rows
) {
   const int child_operator_count = static_cast<int>(non_negated_child_operators.size() + negated_child_operators.size());
   if (updated_number_of_matchers > child_operator_count) {
      return std::make_unique<operators::Empty>(rows);
   }
   if (updated_number_of_matchers < 0) {
      if (match_exactly) {
         return std::make_unique<operators::Empty>(rows);
      }
      return std::make_unique<operators::Full>(rows);
   }
   if (updated_number_of_matchers == 0) {
      if (!match_exactly) {
         return std::make_unique<operators::Full>(rows);
      }
      if (child_operator_count == 0) {
         return std::make_unique<operators::Full>(rows);
      }
      if (child_operator_count == 1) {
         if (non_negated_child_operators.empty()) {
            return std::move(negated_child_operators[0]);
         }
         return std::make_unique<operators::Complement>(std::move(non_negated_child_operators[0]), rows);
      }
      if (negated_child_operators.empty()) {
         auto union_ret = std::make_unique<operators::Union>(std::move(non_negated_child_operators), rows);
         return std::make_unique<operators::Complement>(std::move(union_ret), rows);
      }
      return std::make_unique<operators::Intersection>(std::move(negated_child_operators), std::move(non_negated_child_operators), rows);
   }
   if (updated_number_of_matchers == 1 && child_operator_count == 1) {
      if (negated_child_operators.empty()) {
         return std::move(non_negated_child_operators[0]);
      }
      return std::make_unique<operators::Complement>(std::move(negated_child_operators[0]), rows);
   }
   return nullptr;
}

std::unique_ptr<Operator> toOperator(  // nof.cpp:90-154
   const int updated_number_of_matchers, OperatorVector&& non_negated_child_operators, OperatorVector&& negated_child_operators,
   bool match_exactly, RowSpace rows
) {
   auto tmp = handleTrivialCases(updated_number_of_matchers, non_negated_child_operators, negated_child_operators, match_exactly, rows);
   if (tmp) {
      return tmp;
   }
   const int child_operator_count = static_cast<int>(non_negated_child_operators.size() + negated_child_operators.size());
   if (updated_number_of_matchers == child_operator_count) {  // handleAndCase
      if (non_negated_child_operators.empty()) {
         auto union_ret = std::make_unique<operators::Union>(std::move(negated_child_operators), rows);
         return std::make_unique<operators::Complement>(std::move(union_ret), rows);
      }
      return std::make_unique<operators::Intersection>(std::move(non_negated_child_operators), std::move(negated_child_operators), rows);
   }
   if (updated_number_of_matchers == 1 && !match_exactly) {  // handleOrCase
      if (negated_child_operators.empty()) {
         return std::make_unique<operators::Union>(std::move(non_negated_child_operators), rows);
      }
      auto intersection_ret = std::make_unique<operators::Intersection>(
         std::move(negated_child_operators), std::move(non_negated_child_operators), rows
      );
      return std::make_unique<operators::Complement>(std::move(intersection_ret), rows);
   }
   return std::make_unique<operators::Threshold>(
      std::move(non_negated_child_operators), std::move(negated_child_operators), updated_number_of_matchers, match_exactly, rows
   );
}

}  // namespace

std::string NOf::toString(const Database& database) const {
   std::string res = match_exactly ? "[exactly-" + std::to_string(number_of_matchers) + "-of:" : "[" + std::to_string(number_of_matchers) + "-of:";
   for (const auto& child : children) {
      res += child->toString(database);
      res += ", ";
   }
   return res + "]";
}

std::tuple<OperatorVector, OperatorVector, int> NOf::mapChildExpressions(  // nof.cpp:185-218
   const Database& database, const DatabasePartition& database_partition, AmbiguityMode mode
) const {
   OperatorVector non_negated_child_operators;
   OperatorVector negated_child_operators;
   int updated_number_of_matchers = number_of_matchers;
   for (const auto& child_expression : children) {
      auto child_operator = child_expression->compile(database, database_partition, mode);
      if (child_operator->type() == operators::EMPTY) {
         continue;
      }
      if (child_operator->type() == operators::FULL) {
         updated_number_of_matchers--;
      } else if (child_operator->type() == operators::COMPLEMENT) {
         negated_child_operators.emplace_back(child_operator->negate());
      } else {
         non_negated_child_operators.push_back(std::move(child_operator));
      }
   }
   return {std::move(non_negated_child_operators), std::move(negated_child_operators), updated_number_of_matchers};
}

std::unique_ptr<Operator> NOf::rewriteNonExact(  // nof.cpp:220-258
   const Database& database, const DatabasePartition& database_partition, AmbiguityMode mode
) const {
   const RowSpace rows = rowsOf(database_partition);
   OperatorVector at_least_k;
   {
      auto [non_negated, negated, updated] = mapChildExpressions(database, database_partition, mode);
      at_least_k.emplace_back(toOperator(updated, std::move(non_negated), std::move(negated), false, rows));
   }
   OperatorVector at_least_k_plus_one;
   {
      auto [non_negated, negated, updated] = mapChildExpressions(database, database_partition, mode);
      at_least_k_plus_one.emplace_back(toOperator(updated + 1, std::move(non_negated), std::move(negated), false, rows));
   }
   return toOperator(2, std::move(at_least_k), std::move(at_least_k_plus_one), false, rows);
}

std::unique_ptr<Operator> NOf::compile(const Database& database, const DatabasePartition& database_partition, AmbiguityMode mode) const {
   auto [non_negated_child_operators, negated_child_operators, updated_number_of_matchers] =
      mapChildExpressions(database, database_partition, mode);
   // nof.cpp:268-271
   if (mode != NONE && match_exactly && number_of_matchers < static_cast<int>(children.size())) {
      return rewriteNonExact(database, database_partition, mode);
   }
   return toOperator(
      updated_number_of_matchers, std::move(non_negated_child_operators), std::move(negated_child_operators), match_exactly,
      rowsOf(database_partition)
   );
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
      const std::lock_guard<std::mutex> lock(partition.sparse_cache_mutex);
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
      const std::lock_guard<std::mutex> lock(partition.sparse_cache_mutex);
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
      const std::lock_guard<std::mutex> lock(partition.sparse_cache_mutex);
      const auto [entry, inserted] = partition.sparse_cache.try_emplace(key, std::move(buffer));
      pointer = entry->second.as<uint64_t>();  // another thread may have been first: use its plane, ours returns to the pool
   }
   return std::make_unique<operators::IndexScan>(pointer, rows);
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
   const std::string nuc_sequence_name_or_default = nuc_sequence_name.value_or(database.database_config.default_nucleotide_sequence);
   CHECK_SILO_QUERY(
      database.nuc_sequences.count(nuc_sequence_name_or_default) != 0,
      "Database does not contain the nucleotide sequence with name: '" + nuc_sequence_name_or_default + "'"
   )
   const auto& seq_store_partition = database_partition.nuc_sequences.at(nuc_sequence_name_or_default);
   if (position >= seq_store_partition.reference_sequence.size()) {
      throw QueryParseException(
         "NucleotideEquals position is out of bounds '" + std::to_string(position + 1) + "' > '" +
         std::to_string(seq_store_partition.reference_sequence.size()) + "'"
      );
   }
   const Nucleotide::Symbol nucleotide_symbol = value.value_or(seq_store_partition.reference_sequence.at(position));
   if (mode == UPPER_BOUND) {
      const auto expand = [&]() {  // nucleotide_symbol_equals.cpp:137-149: the symbol or any code that may stand for it
         const auto& symbols_to_match = AMBIGUITY_NUC_SYMBOLS.at(static_cast<uint32_t>(nucleotide_symbol));
         ExpressionVector symbol_filters;
         for (const auto symbol : symbols_to_match) {
            symbol_filters.push_back(std::make_unique<NucleotideSymbolEquals>(nuc_sequence_name_or_default, position, symbol));
         }
         return Or(std::move(symbol_filters)).compile(database, database_partition, NONE);
      };
      return cachedUpperBoundPlane(
         database, database_partition, seq_store_partition.seqstore_id, position, static_cast<uint32_t>(nucleotide_symbol), expand
      );
   }
   return symbolPlane<Nucleotide>(database, seq_store_partition, database_partition, position, nucleotide_symbol);
}

// ---- AASymbolEquals (aa_symbol_equals.cpp:41-92; the ambiguity mode is ignored, :44) ----------------
std::string AASymbolEquals::toString(const Database& /*database*/) const {
   const char symbol_char = value.has_value() ? AminoAcid::symbolToChar(*value) : '.';
   return aa_sequence_name + ":" + std::to_string(position + 1) + std::to_string(symbol_char);
}

std::unique_ptr<Operator> AASymbolEquals::compile(
   const Database& database, const DatabasePartition& database_partition, AmbiguityMode /*mode*/
) const {
   const auto& aa_store_partition = database_partition.aa_sequences.at(aa_sequence_name);  // out_of_range -> 500, as in the reference
   if (position >= aa_store_partition.reference_sequence.size()) {
      throw QueryParseException(
         "AminoAcidEquals position is out of bounds '" + std::to_string(position + 1) + "' > '" +
         std::to_string(aa_store_partition.reference_sequence.size()) + "'"
      );
   }
   const AminoAcid::Symbol aa_symbol = value.value_or(aa_store_partition.reference_sequence.at(position));
   // The reference's rewrite for a deleted STOP symbol recurses forever; the dense
   // store simply returns the intended set.
   return symbolPlane<AminoAcid>(database, aa_store_partition, database_partition, position, aa_symbol);
}

// ---- HasMutation (has_mutation.cpp:35-78) ----------------------------------------------------------
std::string HasMutation::toString(const Database& /*database*/) const {
   const std::string prefix = nuc_sequence_name ? nuc_sequence_name.value() + ":" : "";
   return prefix + std::to_string(position);
}

std::unique_ptr<Operator> HasMutation::compile(const Database& database, const DatabasePartition& database_partition, AmbiguityMode mode) const {
   const std::string nuc_sequence_name_or_default = nuc_sequence_name.value_or(database.database_config.default_nucleotide_sequence);
   CHECK_SILO_QUERY(
      database.nuc_sequences.count(nuc_sequence_name_or_default) != 0,
      "Database does not contain the nucleotide sequence with name: '" + nuc_sequence_name_or_default + "'"
   )
   const Nucleotide::Symbol ref_symbol = database.nuc_sequences.at(nuc_sequence_name_or_default).reference_sequence.at(position);
   if (mode == UPPER_BOUND) {
      auto expression = std::make_unique<Negation>(std::make_unique<NucleotideSymbolEquals>(nuc_sequence_name_or_default, position, ref_symbol));
      return expression->compile(database, database_partition, NONE);
   }
   std::vector<Nucleotide::Symbol> symbols = {NS::A, NS::C, NS::G, NS::T};
   // SILO_COMPAT_REMOVE_QUIRK: the reference calls std::remove without erase (has_mutation.cpp:58-65),
   // so the list keeps four entries; at reference-T positions T itself stays in it.
   (void)std::remove(symbols.begin(), symbols.end(), ref_symbol);
   ExpressionVector symbol_filters;
   for (const auto symbol : symbols) {
      symbol_filters.push_back(std::make_unique<NucleotideSymbolEquals>(nuc_sequence_name_or_default, position, symbol));
   }
   return Or(std::move(symbol_filters)).compile(database, database_partition, NONE);
}

// ---- HasAAMutation (has_aa_mutation.cpp:33-63) -----------------------------------------------------
std::string HasAAMutation::toString(const Database& /*database*/) const {
   return aa_sequence_name + ":" + std::to_string(position);
}

std::unique_ptr<Operator> HasAAMutation::compile(const Database& database, const DatabasePartition& database_partition, AmbiguityMode mode) const {
   const AminoAcid::Symbol ref_symbol = database.aa_sequences.at(aa_sequence_name).reference_sequence.at(position);
   if (mode == UPPER_BOUND) {
      auto expression = std::make_unique<Negation>(std::make_unique<AASymbolEquals>(aa_sequence_name, position, ref_symbol));
      return expression->compile(database, database_partition, NONE);
   }
   std::vector<AminoAcid::Symbol> symbols(AminoAcid::SYMBOLS.begin(), AminoAcid::SYMBOLS.end());
   // SILO_COMPAT_REMOVE_QUIRK: both removes are without erase (has_aa_mutation.cpp:48-52)
   (void)std::remove(symbols.begin(), symbols.end(), AminoAcid::Symbol::X);
   (void)std::remove(symbols.begin(), symbols.end(), ref_symbol);
   ExpressionVector symbol_filters;
   for (const auto symbol : symbols) {
      symbol_filters.push_back(std::make_unique<AASymbolEquals>(aa_sequence_name, position, symbol));
   }
   return Or(std::move(symbol_filters)).compile(database, database_partition, NONE);
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
         const std::lock_guard<std::mutex> lock(database_partition.sparse_cache_mutex);
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
         const std::lock_guard<std::mutex> lock(database_partition.sparse_cache_mutex);
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
