// query_engine.h — host-side mirror of silo::query_engine (same type and member names, same error
// behaviour); the arithmetic behind Operator::evaluate and Action::execute runs in HIP kernels.
//
// Reference interfaces mirrored (file:line under the reference tree):
//   QueryEngine::executeQuery            include/silo/query_engine/query_engine.h:13-23, src/.../query_engine.cpp:30-68
//   Query                                src/silo/query_engine/query.cpp:13-28
//   QueryResult / QueryResultEntry       include/silo/query_engine/query_result.h:14-20, src/.../query_result.cpp:10-25
//   OperatorResult                       include/silo/query_engine/operator_result.h:8-31
//   QueryParseException / CHECK_SILO_QUERY   include/silo/query_engine/query_parse_exception.h:6-16
//   QueryCompilationException            include/silo/query_engine/query_compilation_exception.h
//   operators::Operator + subclasses     include/silo/query_engine/operators/*.h
//   filter_expressions::Expression + subclasses   include/silo/query_engine/filter_expressions/*.h
//   actions::Action, Aggregated, Mutations<S>     include/silo/query_engine/actions/{action,aggregated,mutations}.h
#pragma once

#include <cstdint>
#include <functional>
#include <map>
#include <memory>
#include <optional>
#include <stdexcept>
#include <string>
#include <variant>
#include <vector>

#include "database.h"
#include "json.h"

namespace silo {

class QueryParseException : public std::runtime_error {
  public:
   explicit QueryParseException(const std::string& error_message) : std::runtime_error(error_message) {}
};

class QueryCompilationException : public std::runtime_error {
  public:
   explicit QueryCompilationException(const std::string& error_message) : std::runtime_error(error_message) {}
};

#define CHECK_SILO_QUERY(condition, message)    \
   if (!(condition)) {                          \
      throw silo::QueryParseException(message); \
   }

namespace query_engine {

struct QueryResultEntry {
   std::map<std::string, std::optional<std::variant<std::string, int32_t, double>>> fields;
};

struct QueryResult {
   std::vector<QueryResultEntry> query_result;
};

json::Value toJson(const QueryResult& query_result);  // {"queryResult": [...]}
/// The same document as text, without the intermediate tree.
std::string toJsonText(const QueryResult& query_result);

/// The rows an operator ranges over: the reference passes a bare `row_count`; here it also names
/// the device shard that holds those rows.
struct RowSpace {
   uint32_t row_count = 0;
   const DatabasePartition* partition = nullptr;
};

namespace operators {
class Operator;
}

/// Result of Operator::evaluate.  The reference holds a (borrowed or owned) roaring bitmap; here the
/// evaluation is deferred: cardinality() runs the fused filter kernel in count-only mode, bitset()
/// materialises the row bitset in HBM (both at most once).
class ProgramBuilder;

class OperatorResult {
  public:
   OperatorResult() = default;
   OperatorResult(RowSpace rows, std::unique_ptr<operators::Operator> root);

   [[nodiscard]] uint32_t cardinality() const;
   [[nodiscard]] const uint64_t* bitset() const;  // device pointer, Wp words
   [[nodiscard]] bool isFull() const;              // cardinality() == row_count
   [[nodiscard]] const RowSpace& rows() const;
   /// Runs the kernel now, producing both the bitset and the count in one launch.
   void materialize() const;
   /// Batched counting (QueryEngine::executeQueries): lowers the operator tree into `builder` and hands out the
   /// program, to be launched together with those of other queries (silo_gpu_filter_eval_batch); the caller reports
   /// the cardinality back with setCount.  false: nothing to launch (Empty / Full / already counted).
   [[nodiscard]] bool prepareCount(ProgramBuilder& builder, silo_gpu_bitprog& program) const;
   void setCount(uint32_t count) const;

  private:
   struct State;
   std::shared_ptr<State> state;
};

// ---------------------------------------------------------------------------------------------
// Lowering of an operator tree to the bit-program of include/silo_gpu.h (one fused kernel launch).
// ---------------------------------------------------------------------------------------------
class ProgramBuilder {
  public:
   explicit ProgramBuilder(RowSpace rows) : rows(rows) {}

   uint32_t allocSlot();
   uint32_t allocRun(uint32_t count);
   void freeSlot(uint32_t slot);
   void freeRun(uint32_t slot, uint32_t count);
   void emit(uint32_t op, uint32_t dst, uint32_t a = 0, uint32_t b = 0, uint32_t imm = 0);
   uint32_t leaf(const uint64_t* device_bitset);
   /// Leaf for a symbol that is stored sparsely: scattered into a cached / pooled bitset before launch.
   uint32_t sparseLeaf(uint32_t seqstore_id, uint32_t position, uint32_t symbol);
   const uint64_t* sparsePointer(uint32_t seqstore_id, uint32_t position, uint32_t symbol);
   /// A row bitset from the partition's pool that lives as long as this builder (filled by the caller before run()).
   uint64_t* temporaryBitset();
   /// Appends `columns` as consecutive leaves; returns the imm of an n-ary instruction (first | count << 16).
   uint32_t leafRun(const std::vector<const uint64_t*>& columns);
   /// Children that are stored columns (foldable by one n-ary instruction) vs composite sub-trees.
   struct Split {
      std::vector<const uint64_t*> columns;
      std::vector<const operators::Operator*> composite;
   };
   Split split(const std::vector<std::unique_ptr<operators::Operator>>& children);
   /// Lowers `child` into this program, or — when it would not fit the instruction / leaf budget —
   /// evaluates it with its own launch and loads the result as a leaf.
   uint32_t lowerChild(const operators::Operator& child);

   /// Launches the program (result expected in `result_slot`).
   void run(uint32_t result_slot, uint64_t* out_bitset, uint64_t* out_count, void* stream);
   /// Launches the program and returns the cardinality of its result, delivered by the kernel itself through this
   /// thread's count slot (no counter memset, no device-to-host copy, no stream synchronisation).
   uint32_t runCounting(uint32_t result_slot, uint64_t* out_bitset, void* stream);

   /// Lowering queued device work of its own on the lowering thread's stream (metadata predicates, insertion searches
   /// write temporary bitsets): a program that is launched from ANOTHER stream has to wait for that work first.
   [[nodiscard]] bool queuedDeviceWork() const { return !temporaries.empty(); }
   /// The finished program (result expected in `result_slot`); points into this builder, which must outlive its use.
   silo_gpu_bitprog finishProgram(uint32_t result_slot);

   const RowSpace rows;

  private:
   std::vector<uint32_t> code;
   std::vector<const uint64_t*> leaves;
   std::vector<DeviceBuffer> temporaries;
   std::vector<OperatorResult> materialized_children;
   uint32_t used_slots = 0;  // bit mask
   uint32_t high_water = 0;
};

namespace operators {

enum Type {
   EMPTY, FULL, INDEX_SCAN, INTERSECTION, COMPLEMENT, RANGE_SELECTION, SELECTION, BITMAP_SELECTION, THRESHOLD, UNION, BITMAP_PRODUCER
};

struct Cost {
   uint32_t instructions = 0;
   uint32_t leaves = 0;
};

class Operator {
  public:
   explicit Operator(RowSpace rows) : rows(rows) {}
   virtual ~Operator() noexcept = default;

   [[nodiscard]] virtual Type type() const = 0;
   virtual OperatorResult evaluate() const;
   /// The same for a tree nobody else needs any more: the result takes the tree over instead of copying it.
   static OperatorResult evaluate(std::unique_ptr<Operator> root);
   virtual std::string toString() const = 0;
   virtual std::unique_ptr<Operator> copy() const = 0;
   virtual std::unique_ptr<Operator> negate() const = 0;

   /// Emits this subtree into `builder`; returns the slot (or leaf operand) that holds its value.
   virtual uint32_t lower(ProgramBuilder& builder) const = 0;
   /// Device pointer when this operator is just a stored column (IndexScan, BitmapSelection CONTAINS).
   virtual const uint64_t* storedColumn(ProgramBuilder& /*builder*/) const { return nullptr; }
   [[nodiscard]] virtual Cost cost() const = 0;

   const RowSpace rows;
};

using OperatorVector = std::vector<std::unique_ptr<Operator>>;

class Empty : public Operator {
  public:
   explicit Empty(RowSpace rows) : Operator(rows) {}
   Type type() const override { return EMPTY; }
   std::string toString() const override { return "Empty"; }
   std::unique_ptr<Operator> copy() const override;
   std::unique_ptr<Operator> negate() const override;
   uint32_t lower(ProgramBuilder& builder) const override;
   Cost cost() const override { return {1, 0}; }
};

class Full : public Operator {
  public:
   explicit Full(RowSpace rows) : Operator(rows) {}
   Type type() const override { return FULL; }
   std::string toString() const override { return "Full"; }
   std::unique_ptr<Operator> copy() const override;
   std::unique_ptr<Operator> negate() const override;
   uint32_t lower(ProgramBuilder& builder) const override;
   Cost cost() const override { return {1, 0}; }
};

/// index_scan.cpp: borrows a stored bitmap; here a dense plane pointer in HBM, or a sparse symbol.
class IndexScan : public Operator {
  public:
   IndexScan(const uint64_t* bitmap, RowSpace rows) : Operator(rows), bitmap(bitmap) {}
   IndexScan(uint32_t seqstore_id, uint32_t position, uint32_t symbol, RowSpace rows)
       : Operator(rows), sparse(true), seqstore_id(seqstore_id), position(position), symbol(symbol) {}
   Type type() const override { return INDEX_SCAN; }
   std::string toString() const override { return "IndexScan"; }
   std::unique_ptr<Operator> copy() const override;
   std::unique_ptr<Operator> negate() const override;
   uint32_t lower(ProgramBuilder& builder) const override;
   const uint64_t* storedColumn(ProgramBuilder& builder) const override;
   Cost cost() const override { return {1, 1}; }

   const uint64_t* bitmap = nullptr;
   bool sparse = false;
   uint32_t seqstore_id = 0, position = 0, symbol = 0;
   /// Owner of `bitmap` when it is a plane received from another rank (position-range sharding).
   std::shared_ptr<DeviceBuffer> received;
};

/// bitmap_selection.cpp probes the row-wise missing-symbol bitmaps; the dense store keeps the missing
/// symbol as a column plane, so CONTAINS is that plane and NOT_CONTAINS its complement.
class BitmapSelection : public Operator {
  public:
   enum Comparator { CONTAINS, NOT_CONTAINS };
   BitmapSelection(const uint64_t* missing_plane, RowSpace rows, Comparator comparator, uint32_t value)
       : Operator(rows), missing_plane(missing_plane), comparator(comparator), value(value) {}
   /// The missing symbol of a store that keeps it as runs (no resident plane): the position's plane is materialised, and
   /// cached, when the operator is lowered — like a sparsely stored symbol of an IndexScan.
   BitmapSelection(uint32_t seqstore_id, uint32_t local_position, uint32_t symbol, RowSpace rows, Comparator comparator, uint32_t value)
       : Operator(rows), missing_plane(nullptr), comparator(comparator), value(value), materialise(true), seqstore_id(seqstore_id),
         local_position(local_position), symbol(symbol) {}
   Type type() const override { return BITMAP_SELECTION; }
   std::string toString() const override { return "BitmapSelection"; }
   std::unique_ptr<Operator> copy() const override;
   std::unique_ptr<Operator> negate() const override;
   uint32_t lower(ProgramBuilder& builder) const override;
   const uint64_t* storedColumn(ProgramBuilder& builder) const override;
   Cost cost() const override { return {2, 1}; }

   const uint64_t* missing_plane;
   Comparator comparator;
   uint32_t value;  // the position, kept for parity with the reference's constructor
   bool materialise = false;
   uint32_t seqstore_id = 0;
   uint32_t local_position = 0;
   uint32_t symbol = 0;
};

class Complement : public Operator {
  public:
   Complement(std::unique_ptr<Operator> child, RowSpace rows) : Operator(rows), child(std::move(child)) {}
   Type type() const override { return COMPLEMENT; }
   std::string toString() const override { return "!" + child->toString(); }
   std::unique_ptr<Operator> copy() const override;
   std::unique_ptr<Operator> negate() const override;
   uint32_t lower(ProgramBuilder& builder) const override;
   Cost cost() const override;

   std::unique_ptr<Operator> child;
};

class Intersection : public Operator {
  public:
   Intersection(OperatorVector&& children, OperatorVector&& negated_children, RowSpace rows);
   Type type() const override { return INTERSECTION; }
   std::string toString() const override;
   std::unique_ptr<Operator> copy() const override;
   std::unique_ptr<Operator> negate() const override;
   uint32_t lower(ProgramBuilder& builder) const override;
   Cost cost() const override;

   OperatorVector children;
   OperatorVector negated_children;
};

class Union : public Operator {
  public:
   Union(OperatorVector&& children, RowSpace rows) : Operator(rows), children(std::move(children)) {}
   Type type() const override { return UNION; }
   std::string toString() const override;
   std::unique_ptr<Operator> copy() const override;
   std::unique_ptr<Operator> negate() const override;
   uint32_t lower(ProgramBuilder& builder) const override;
   Cost cost() const override;

   OperatorVector children;
};

/// One comparison of a metadata column with a constant (CompareToValueSelection<T>, selection.h:52-70): on the
/// device a k_bitset_from_compare launch that turns the column into a row bitset.
struct Predicate {
   const storage::column::MetadataColumnPartition* column;
   int comparator;  // SILO_GPU_CMP_*
   union {
      int32_t as_int;
      uint32_t as_word;
      double as_double;
   } value;
   [[nodiscard]] Predicate negated() const;  // the comparator is negated, selection.cpp:195-220
};

/// selection.cpp: rows (of `child`, or all) that satisfy every predicate.
class Selection : public Operator {
  public:
   Selection(std::unique_ptr<Operator> child, std::vector<Predicate> predicates, RowSpace rows)
       : Operator(rows), child(std::move(child)), predicates(std::move(predicates)) {}
   Selection(std::vector<Predicate> predicates, RowSpace rows) : Operator(rows), predicates(std::move(predicates)) {}
   Type type() const override { return SELECTION; }
   std::string toString() const override { return "Select[" + std::to_string(predicates.size()) + " predicates]"; }
   std::unique_ptr<Operator> copy() const override;
   std::unique_ptr<Operator> negate() const override;
   uint32_t lower(ProgramBuilder& builder) const override;
   Cost cost() const override;

   std::unique_ptr<Operator> child;  // may be null
   std::vector<Predicate> predicates;
};

/// bitmap_producer.cpp: a bitmap computed at evaluation time.  The only producer on this path is the insertion
/// search (insertion_contains.cpp:104-113): the rows that carry one of the distinct insertions the pattern matched,
/// scattered into a bitset by k_bitset_from_pairs.
class BitmapProducer : public Operator {
  public:
   BitmapProducer(const storage::column::InsertionColumnPartition::SequenceIndex* index, std::vector<uint8_t> membership, RowSpace rows)
       : Operator(rows), index(index), membership(std::move(membership)) {}
   Type type() const override { return BITMAP_PRODUCER; }
   std::string toString() const override { return "BitmapProducer"; }
   std::unique_ptr<Operator> copy() const override;
   std::unique_ptr<Operator> negate() const override;
   uint32_t lower(ProgramBuilder& builder) const override;
   Cost cost() const override { return {1, 1}; }

   const storage::column::InsertionColumnPartition::SequenceIndex* index;
   std::vector<uint8_t> membership;  // per distinct insertion id of the index
};

class Threshold : public Operator {
  public:
   Threshold(OperatorVector&& non_negated_children, OperatorVector&& negated_children, uint32_t number_of_matchers, bool match_exactly, RowSpace rows);
   Type type() const override { return THRESHOLD; }
   std::string toString() const override;
   std::unique_ptr<Operator> copy() const override;
   std::unique_ptr<Operator> negate() const override;
   uint32_t lower(ProgramBuilder& builder) const override;
   Cost cost() const override;

   OperatorVector non_negated_children;
   OperatorVector negated_children;
   uint32_t number_of_matchers;
   bool match_exactly;
};

}  // namespace operators

namespace filter_expressions {

struct Expression {
   enum AmbiguityMode { UPPER_BOUND, LOWER_BOUND, NONE };
   virtual ~Expression() = default;
   virtual std::string toString(const Database& database) const = 0;
   [[nodiscard]] virtual std::unique_ptr<operators::Operator> compile(
      const Database& database, const DatabasePartition& database_partition, AmbiguityMode mode
   ) const = 0;
};

Expression::AmbiguityMode invertMode(Expression::AmbiguityMode mode);

/// expression.cpp:49-102
std::unique_ptr<Expression> parseExpression(const json::Value& json);

using ExpressionVector = std::vector<std::unique_ptr<Expression>>;

#define SILO_DECLARE_EXPRESSION_METHODS                                        \
   std::string toString(const Database& database) const override;            \
   [[nodiscard]] std::unique_ptr<operators::Operator> compile(                 \
      const Database& database, const DatabasePartition& database_partition, AmbiguityMode mode \
   ) const override;

struct True : public Expression {
   SILO_DECLARE_EXPRESSION_METHODS
};
struct False : public Expression {
   SILO_DECLARE_EXPRESSION_METHODS
};
struct And : public Expression {
   explicit And(ExpressionVector&& children) : children(std::move(children)) {}
   SILO_DECLARE_EXPRESSION_METHODS
   ExpressionVector children;
};
struct Or : public Expression {
   explicit Or(ExpressionVector&& children) : children(std::move(children)) {}
   SILO_DECLARE_EXPRESSION_METHODS
   ExpressionVector children;
};
struct Negation : public Expression {
   explicit Negation(std::unique_ptr<Expression> child) : child(std::move(child)) {}
   SILO_DECLARE_EXPRESSION_METHODS
   std::unique_ptr<Expression> child;
};
struct Maybe : public Expression {
   explicit Maybe(std::unique_ptr<Expression> child) : child(std::move(child)) {}
   SILO_DECLARE_EXPRESSION_METHODS
   std::unique_ptr<Expression> child;
};
struct Exact : public Expression {
   explicit Exact(std::unique_ptr<Expression> child) : child(std::move(child)) {}
   SILO_DECLARE_EXPRESSION_METHODS
   std::unique_ptr<Expression> child;
};
struct NOf : public Expression {
   NOf(ExpressionVector&& children, int number_of_matchers, bool match_exactly)
       : children(std::move(children)), number_of_matchers(number_of_matchers), match_exactly(match_exactly) {}
   SILO_DECLARE_EXPRESSION_METHODS
   ExpressionVector children;
   int number_of_matchers;
   bool match_exactly;
};
struct NucleotideSymbolEquals : public Expression {
   NucleotideSymbolEquals(std::optional<std::string> nuc_sequence_name, uint32_t position, std::optional<Nucleotide::Symbol> value)
       : nuc_sequence_name(std::move(nuc_sequence_name)), position(position), value(value) {}
   SILO_DECLARE_EXPRESSION_METHODS
   std::optional<std::string> nuc_sequence_name;
   uint32_t position;
   std::optional<Nucleotide::Symbol> value;
};
struct AASymbolEquals : public Expression {
   AASymbolEquals(std::string aa_sequence_name, uint32_t position, std::optional<AminoAcid::Symbol> value)
       : aa_sequence_name(std::move(aa_sequence_name)), position(position), value(value) {}
   SILO_DECLARE_EXPRESSION_METHODS
   std::string aa_sequence_name;
   uint32_t position;
   std::optional<AminoAcid::Symbol> value;
};
struct HasMutation : public Expression {
   HasMutation(std::optional<std::string> nuc_sequence_name, uint32_t position)
       : nuc_sequence_name(std::move(nuc_sequence_name)), position(position) {}
   SILO_DECLARE_EXPRESSION_METHODS
   std::optional<std::string> nuc_sequence_name;
   uint32_t position;
};
struct HasAAMutation : public Expression {
   HasAAMutation(std::string aa_sequence_name, uint32_t position)
       : aa_sequence_name(std::move(aa_sequence_name)), position(position) {}
   SILO_DECLARE_EXPRESSION_METHODS
   std::string aa_sequence_name;
   uint32_t position;
};
struct PangoLineageFilter : public Expression {
   PangoLineageFilter(std::string column, std::string lineage, bool include_sublineages)
       : column(std::move(column)), lineage(std::move(lineage)), include_sublineages(include_sublineages) {}
   SILO_DECLARE_EXPRESSION_METHODS
   std::string column;
   std::string lineage;
   bool include_sublineages;
};
// ---- metadata predicates ----
struct StringEquals : public Expression {  // string_equals.cpp
   StringEquals(std::string column, std::string value) : column(std::move(column)), value(std::move(value)) {}
   SILO_DECLARE_EXPRESSION_METHODS
   std::string column;
   std::string value;
};
struct IntEquals : public Expression {  // int_equals.cpp
   IntEquals(std::string column, int32_t value) : column(std::move(column)), value(value) {}
   SILO_DECLARE_EXPRESSION_METHODS
   std::string column;
   int32_t value;
};
struct IntBetween : public Expression {  // int_between.cpp
   IntBetween(std::string column, std::optional<int32_t> from, std::optional<int32_t> to) : column(std::move(column)), from(from), to(to) {}
   SILO_DECLARE_EXPRESSION_METHODS
   std::string column;
   std::optional<int32_t> from;
   std::optional<int32_t> to;
};
struct FloatEquals : public Expression {  // float_equals.cpp
   FloatEquals(std::string column, double value) : column(std::move(column)), value(value) {}
   SILO_DECLARE_EXPRESSION_METHODS
   std::string column;
   double value;
};
struct FloatBetween : public Expression {  // float_between.cpp
   FloatBetween(std::string column, std::optional<double> from, std::optional<double> to) : column(std::move(column)), from(from), to(to) {}
   SILO_DECLARE_EXPRESSION_METHODS
   std::string column;
   std::optional<double> from;
   std::optional<double> to;
};
template <typename SymbolType>
struct InsertionContains : public Expression {  // insertion_contains.cpp
   InsertionContains(std::vector<std::string>&& column_names, std::optional<std::string> sequence_name, uint32_t position, std::string value)
       : column_names(std::move(column_names)), sequence_name(std::move(sequence_name)), position(position), value(std::move(value)) {}
   SILO_DECLARE_EXPRESSION_METHODS
   std::vector<std::string> column_names;
   std::optional<std::string> sequence_name;
   uint32_t position;
   std::string value;
};
struct DateBetween : public Expression {  // date_between.cpp
   DateBetween(std::string column, std::optional<common::Date> date_from, std::optional<common::Date> date_to)
       : column(std::move(column)), date_from(date_from), date_to(date_to) {}
   SILO_DECLARE_EXPRESSION_METHODS
   std::string column;
   std::optional<common::Date> date_from;
   std::optional<common::Date> date_to;
};

}  // namespace filter_expressions

namespace actions {

struct OrderByField {
   std::string name;
   bool ascending;
};

/// Scans of several queries that read the same planes are collected here and issued as batched launches
/// (silo_gpu_mutations_scan_batch: one pass over the planes for up to 4 filters).  A batcher is active on a
/// thread only inside QueryEngine::executeQueries; otherwise every scan is launched at once.
class ScanBatcher {
  public:
   struct Request {
      silo_gpu_store* store;
      uint32_t seqstore_id;
      const uint64_t* filter;  // nullptr = full filter
      uint32_t pos_begin, pos_end;
      uint32_t* counts;
   };
   ScanBatcher();
   ~ScanBatcher();
   ScanBatcher(const ScanBatcher&) = delete;
   static ScanBatcher* active();
   void add(const Request& request) { requests.push_back(request); }
   /// With collectives installed: sum `n` counts across ranks after the scans of the batch were launched
   /// (every rank runs the same batch, so the reductions pair up in recording order).
   void addReduction(const Database& database, uint32_t* device_values, size_t n) { reductions.push_back({&database, device_values, n}); }
   /// Work that has to follow the launches (and reductions) of the batch on the stream, e.g. fetching the results.
   void afterFlush(std::function<void()> callback) { after_flush.push_back(std::move(callback)); }
   /// What a query recorded can be dropped again when it fails before the flush (its buffers die with it).
   struct Checkpoint {
      size_t requests, reductions, after_flush;
   };
   [[nodiscard]] Checkpoint checkpoint() const { return {requests.size(), reductions.size(), after_flush.size()}; }
   void rollback(const Checkpoint& mark) {
      requests.resize(mark.requests);
      reductions.resize(mark.reductions);
      after_flush.resize(mark.after_flush);
   }
   void flush();

  private:
   struct Reduction {
      const Database* database;
      uint32_t* device_values;
      size_t n;
   };
   std::vector<Request> requests;
   std::vector<Reduction> reductions;
   std::vector<std::function<void()>> after_flush;
   ScanBatcher* previous;
};

class Action {
  protected:
   std::vector<OrderByField> order_by_fields;
   std::optional<uint32_t> limit;
   std::optional<uint32_t> offset;

   void applySort(QueryResult& result) const;
   void applyOffsetAndLimit(QueryResult& result) const;
   virtual void validateOrderByFields(const Database& database) const = 0;
   [[nodiscard]] virtual QueryResult execute(const Database& database, std::vector<OperatorResult> bitmap_filter) const = 0;

  public:
   /// State of an action between its two phases (see begin / finish).
   struct Pending {
      virtual ~Pending() = default;
      std::vector<OperatorResult> bitmap_filter;
   };

   virtual ~Action() = default;
   /// The action needs nothing but the cardinality of the filter (Aggregated without groupByFields): a batch of such
   /// queries evaluates all its filters in one launch.
   [[nodiscard]] virtual bool countsOnly() const { return false; }
   void setOrdering(const std::vector<OrderByField>& order_by_fields, std::optional<uint32_t> limit, std::optional<uint32_t> offset);
   [[nodiscard]] virtual QueryResult executeAndOrder(const Database& database, std::vector<OperatorResult> bitmap_filter) const;

   /// Two-phase form of executeAndOrder for batches of queries: begin() validates and queues the device
   /// work (with a ScanBatcher active the scans are only recorded), finish() fetches the results, builds the
   /// rows and applies ordering / offset / limit.  The default runs everything in finish().
   [[nodiscard]] virtual std::unique_ptr<Pending> begin(const Database& database, std::vector<OperatorResult> bitmap_filter) const;
   [[nodiscard]] virtual QueryResult finish(const Database& database, Pending& pending) const;
   /// finish() as the response body — the bytes of toJsonText(finish(...)).  An action whose rows come off the device as
   /// plain numbers (Mutations) writes them straight into the text: the map-of-variants rows the reference's QueryResult
   /// prescribes (query_result.h:14-20) cost more to build than the whole scan of a small query takes.
   [[nodiscard]] virtual std::string finishJson(const Database& database, Pending& pending) const;

  protected:
   [[nodiscard]] QueryResult orderAndLimit(QueryResult result) const;
};

/// action.cpp:144-187
std::unique_ptr<Action> parseAction(const json::Value& json);

class Aggregated : public Action {
   std::vector<std::string> group_by_fields;
   void validateOrderByFields(const Database& database) const override;
   [[nodiscard]] QueryResult execute(const Database& database, std::vector<OperatorResult> bitmap_filter) const override;
   /// aggregated.cpp:100-149 — per partition one k_group_count launch over the dictionary ids of the fields
   [[nodiscard]] QueryResult aggregateWithGrouping(const Database& database, std::vector<OperatorResult>& bitmap_filter) const;

  public:
   explicit Aggregated(std::vector<std::string> group_by_fields) : group_by_fields(std::move(group_by_fields)) {}
   [[nodiscard]] bool countsOnly() const override { return group_by_fields.empty(); }
};

/// details.cpp: the metadata of the selected rows.  The filter runs on the device; the rows are read from the host
/// copies of the columns.
class Details : public Action {
   std::vector<std::string> fields;
   void validateOrderByFields(const Database& database) const override;
   [[nodiscard]] QueryResult execute(const Database& database, std::vector<OperatorResult> bitmap_filter) const override;

  public:
   explicit Details(std::vector<std::string> fields) : fields(std::move(fields)) {}
   [[nodiscard]] QueryResult executeAndOrder(const Database& database, std::vector<OperatorResult> bitmap_filter) const override;
   [[nodiscard]] QueryResult finish(const Database& database, Pending& pending) const override;
};

/// fasta.cpp: primary key + the unaligned nucleotide sequences of the selected rows (host data only).
class Fasta : public Action {
   std::vector<std::string> sequence_names;
   void validateOrderByFields(const Database& database) const override;
   [[nodiscard]] QueryResult execute(const Database& database, std::vector<OperatorResult> bitmap_filter) const override;

  public:
   static constexpr size_t SEQUENCE_LIMIT = 10'000;
   explicit Fasta(std::vector<std::string>&& sequence_names) : sequence_names(std::move(sequence_names)) {}
};

/// insertions.cpp: the distinct insertions of the selected rows with their counts (k_count_pairs per insertion index).
template <typename SymbolType>
class InsertionAggregation : public Action {
   std::vector<std::string> column_names;
   std::vector<std::string> sequence_names;
   void validateOrderByFields(const Database& database) const override;
   [[nodiscard]] QueryResult execute(const Database& database, std::vector<OperatorResult> bitmap_filter) const override;

  public:
   InsertionAggregation(std::vector<std::string>&& column_names, std::vector<std::string>&& sequence_names)
       : column_names(std::move(column_names)), sequence_names(std::move(sequence_names)) {}
};

/// fasta_aligned.cpp: primary key + the aligned sequences of the selected rows, gathered from the planes on the
/// device (k_reconstruct_sequences).
class FastaAligned : public Action {
   std::vector<std::string> sequence_names;
   void validateOrderByFields(const Database& database) const override;
   [[nodiscard]] QueryResult execute(const Database& database, std::vector<OperatorResult> bitmap_filter) const override;

  public:
   explicit FastaAligned(std::vector<std::string>&& sequence_names) : sequence_names(std::move(sequence_names)) {}
};

template <typename SymbolType>
class Mutations : public Action {
   std::vector<std::string> sequence_names;
   double min_proportion;

   const std::string MUTATION_FIELD_NAME = "mutation";
   const std::string SEQUENCE_FIELD_NAME = "sequenceName";
   const std::string PROPORTION_FIELD_NAME = "proportion";
   const std::string COUNT_FIELD_NAME = "count";

   struct PrefilteredBitmaps {
      std::vector<std::pair<const OperatorResult&, const SequenceStorePartition<SymbolType>&>> bitmaps;
      std::vector<std::pair<const OperatorResult&, const SequenceStorePartition<SymbolType>&>> full_bitmaps;
   };

   static std::map<std::string, PrefilteredBitmaps> preFilterBitmaps(const Database& database, std::vector<OperatorResult>& bitmap_filter);

   /// Launches the K1 scans of one sequence store into its slice counts[position][valid symbol] of the query's
   /// count table (accumulating over partitions); returns with the scans in flight on this thread's stream.
   static void calculateMutationsPerPosition(
      const Database& database, const SequenceStore<SymbolType>& sequence_store, const PrefilteredBitmaps& bitmap_filter,
      uint32_t* device_counts
   );

   /// `counts` = the slice counts[position][valid symbol] of one store, on the host.
   void addMutationsToOutput(
      const std::string& sequence_name, const SequenceStore<SymbolType>& sequence_store, const uint32_t* counts,
      std::vector<QueryResultEntry>& output
   ) const;

   void validateOrderByFields(const Database& database) const override;
   [[nodiscard]] QueryResult execute(const Database& database, std::vector<OperatorResult> bitmap_filter) const override;

   /// One row of the result from a cell the device selected.
   void addSelectedRowToOutput(
      const std::string& sequence_name, const SequenceStore<SymbolType>& sequence_store, uint32_t position, const silo_gpu_mutation_row& row,
      std::vector<QueryResultEntry>& output
   ) const;
   struct PendingScans : public Action::Pending {
      std::vector<std::string> sequence_names;  // the requested stores, in output order
      /// counts of all stores of the alphabet (MutationTableLayout), then the list written by k_mutations_select:
      /// [n_rows, -, -, -] + row_capacity rows
      DeviceBuffer device_table;
      size_t table_bytes = 0;
      uint32_t row_capacity = 0;  // 0: the host selects from the whole table
      HostFetch fetch;            // the whole table, enqueued right behind the scans (row_capacity == 0)
      RowSlot row_slot;           // the list of selected rows, written by the kernel straight into page-locked host memory
      HostFetch check_fetch;      // sharded: the two fingerprint sums behind the table (checkSameQuery)
      uint64_t fingerprint = 0;
   };
   [[nodiscard]] QueryResult collect(const Database& database, PendingScans& scans) const;
   /// A result row as the device (or the host's pass over the whole table) selects it, before any text is made.
   struct SelectedRow {
      const std::string* sequence_name;
      const SequenceStore<SymbolType>* sequence_store;
      uint32_t position;  // within the store
      uint32_t symbol_index;
      uint32_t count;
      uint32_t total;
   };
   /// The selected rows in the reference's output order: stores in request order, positions ascending, symbols in
   /// VALID_MUTATION_SYMBOLS order (mutations.cpp:190-229, 259-268).
   [[nodiscard]] std::vector<SelectedRow> collectSelected(const Database& database, PendingScans& scans) const;
   [[nodiscard]] std::string mutationName(const SelectedRow& row) const;

  public:
   [[nodiscard]] std::unique_ptr<Action::Pending> begin(const Database& database, std::vector<OperatorResult> bitmap_filter) const override;
   [[nodiscard]] QueryResult finish(const Database& database, Action::Pending& pending) const override;
   [[nodiscard]] std::string finishJson(const Database& database, Action::Pending& pending) const override;

   Mutations(std::vector<std::string>&& sequence_names, double min_proportion)
       : sequence_names(std::move(sequence_names)), min_proportion(min_proportion) {}
};

}  // namespace actions

class Query {
  public:
   std::unique_ptr<filter_expressions::Expression> filter;
   std::unique_ptr<actions::Action> action;
   explicit Query(const std::string& query_string);
};

class QueryEngine {
  private:
   const Database& database;

  public:
   explicit QueryEngine(const Database& database) : database(database) {}
   virtual ~QueryEngine() = default;
   [[nodiscard]] virtual QueryResult executeQuery(const std::string& query) const;
   /// executeQuery as the response body (the bytes of toJsonText(executeQuery(query))): what a caller that only forwards
   /// the JSON — silo_api's QueryHandler::post, query_handler.cpp:38-41 — needs, without the QueryResult rows in between.
   [[nodiscard]] std::string executeQueryJson(const std::string& query) const;

   /// Outcome of one query of a batch: a result, or the exception it raised (mapped to 400 / 500 by the caller
   /// exactly as for a single query, silo_api/query_handler.cpp:42-73).
   struct BatchOutcome {
      QueryResult result;
      std::string json;  // the response body, when executeQueries was asked to render it (while later queries' scans still run)
      std::exception_ptr error;
   };
   /// Executes a batch of independent queries: filters are evaluated per query, then the Mutations scans of all
   /// queries that read the same sequence store are issued together, several filters per pass over the planes.
   [[nodiscard]] std::vector<BatchOutcome> executeQueries(const std::vector<std::string>& queries, bool render_json = false) const;
};

}  // namespace query_engine
}  // namespace silo
