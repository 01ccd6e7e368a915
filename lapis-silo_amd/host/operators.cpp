// operators.cpp — physical operators and their lowering to the fused bit-program (K3).
// Reference: src/silo/query_engine/operators/*.cpp, src/silo/query_engine/operator_result.cpp.
#include <algorithm>
#include <bit>
#include <mutex>
#include <shared_mutex>

#include "query_engine.h"

namespace silo::query_engine {

// ---- OperatorResult ----------------------------------------------------------------------------
struct OperatorResult::State {
   RowSpace rows;
   std::unique_ptr<operators::Operator> root;
   std::mutex mutex;
   DeviceBuffer bitset;
   std::optional<uint32_t> count;
   const uint64_t* borrowed = nullptr;  // IndexScan roots need no kernel for the bitset
};

OperatorResult::OperatorResult(RowSpace rows, std::unique_ptr<operators::Operator> root) : state(std::make_shared<State>()) {
   state->rows = rows;
   state->root = std::move(root);
}

const RowSpace& OperatorResult::rows() const {
   return state->rows;
}

namespace {

constexpr size_t COUNT_BYTES = SILO_GPU_COUNT_SHARDS * sizeof(uint64_t);

DeviceBuffer zeroedCounter(const DatabasePartition& partition) {
   DeviceBuffer counter = partition.pool.acquire(COUNT_BYTES);
   checkGpu(silo_gpu_memset_async(counter.get(), 0, COUNT_BYTES, queryStream()), "silo_gpu_memset_async");
   return counter;
}

uint32_t readCount(const DeviceBuffer& counter) {
   uint64_t shards[SILO_GPU_COUNT_SHARDS];
   checkGpu(silo_gpu_memcpy_d2h(shards, counter.get(), COUNT_BYTES, queryStream()), "silo_gpu_memcpy_d2h");
   uint64_t total = 0;
   for (const uint64_t shard : shards) {
      total += shard;
   }
   return static_cast<uint32_t>(total);
}

bool isLeafOperand(uint32_t operand) {
   return operand >= SILO_GPU_LEAF_OPERAND;
}

}  // namespace

void OperatorResult::materialize() const {
   const std::lock_guard<std::mutex> lock(state->mutex);
   if ((state->bitset || state->borrowed != nullptr) && state->count.has_value()) {
      return;
   }
   const DatabasePartition& partition = *state->rows.partition;
   const size_t row_bytes = static_cast<size_t>(partition.rowWords()) * sizeof(uint64_t);
   const auto* scan = state->root->type() == operators::INDEX_SCAN ? dynamic_cast<const operators::IndexScan*>(state->root.get()) : nullptr;
   if (scan != nullptr && !scan->sparse) {
      state->borrowed = scan->bitmap;  // index_scan.cpp:28-30: borrow, no copy
      if (scan->received == nullptr) {  // a stored bitmap (not a plane received from another rank): its cardinality is kept
         const std::shared_lock<std::shared_mutex> lock(partition.cardinality_cache_mutex);
         const auto known = partition.cardinality_cache.find(scan->bitmap);
         if (known != partition.cardinality_cache.end()) {
            state->count = known->second;
            return;
         }
      }
      DeviceBuffer counter = zeroedCounter(partition);
      checkGpu(silo_gpu_popcount(partition.store, scan->bitmap, counter.as<uint64_t>(), queryStream()), "silo_gpu_popcount");
      state->count = readCount(counter);
      if (scan->received == nullptr) {
         const std::unique_lock<std::shared_mutex> lock(partition.cardinality_cache_mutex);
         if (partition.cardinality_cache.size() < (size_t{1} << 20)) {
            partition.cardinality_cache.emplace(scan->bitmap, *state->count);
         }
      }
   } else {
      DeviceBuffer out = partition.pool.acquire(row_bytes);
      ProgramBuilder builder(state->rows);
      const uint32_t slot = state->root->lower(builder);
      state->count = builder.runCounting(slot, out.as<uint64_t>(), queryStream());
      state->bitset = std::move(out);
   }
}

uint32_t OperatorResult::cardinality() const {
   {
      const std::lock_guard<std::mutex> lock(state->mutex);
      if (state->count.has_value()) {
         return *state->count;
      }
      // count-only launch: no bitset is written (Aggregated never needs it)
      if (state->root->type() == operators::EMPTY) {
         state->count = 0;
         return 0;
      }
      if (state->root->type() == operators::FULL) {
         state->count = state->rows.row_count;
         return *state->count;
      }
      const auto* scan = state->root->type() == operators::INDEX_SCAN ? dynamic_cast<const operators::IndexScan*>(state->root.get()) : nullptr;
      const bool stored = scan != nullptr && !scan->sparse && scan->received == nullptr;
      const DatabasePartition& partition = *state->rows.partition;
      if (stored) {  // the kept cardinality of a stored bitmap (see DatabasePartition::cardinality_cache)
         const std::shared_lock<std::shared_mutex> lock(partition.cardinality_cache_mutex);
         const auto known = partition.cardinality_cache.find(scan->bitmap);
         if (known != partition.cardinality_cache.end()) {
            state->count = known->second;
            return *state->count;
         }
      }
      ProgramBuilder builder(state->rows);
      const uint32_t slot = state->root->lower(builder);
      state->count = builder.runCounting(slot, nullptr, queryStream());
      if (stored) {
         const std::unique_lock<std::shared_mutex> lock(partition.cardinality_cache_mutex);
         if (partition.cardinality_cache.size() < (size_t{1} << 20)) {
            partition.cardinality_cache.emplace(scan->bitmap, *state->count);
         }
      }
      return *state->count;
   }
}

bool OperatorResult::prepareCount(ProgramBuilder& builder, silo_gpu_bitprog& program) const {
   const std::lock_guard<std::mutex> lock(state->mutex);
   if (state->count.has_value()) {
      return false;
   }
   if (state->root->type() == operators::EMPTY || state->root->type() == operators::FULL) {
      state->count = state->root->type() == operators::EMPTY ? 0 : state->rows.row_count;
      return false;
   }
   program = builder.finishProgram(state->root->lower(builder));
   return true;
}

void OperatorResult::setCount(uint32_t count) const {
   const std::lock_guard<std::mutex> lock(state->mutex);
   state->count = count;
}

const uint64_t* OperatorResult::bitset() const {
   materialize();
   return state->borrowed != nullptr ? state->borrowed : state->bitset.as<uint64_t>();
}

bool OperatorResult::isFull() const {
   return cardinality() == state->rows.row_count;
}

// ---- ProgramBuilder ----------------------------------------------------------------------------
uint32_t ProgramBuilder::allocRun(uint32_t count) {
   for (uint32_t first = 0; first + count <= SILO_GPU_MAX_SLOTS; ++first) {
      const uint32_t mask = (count == 32 ? 0xFFFFFFFFu : ((1u << count) - 1u)) << first;
      if ((used_slots & mask) == 0) {
         used_slots |= mask;
         high_water = std::max(high_water, first + count);
         return first;
      }
   }
   throw QueryCompilationException("Compilation Error: filter expression needs more than " + std::to_string(SILO_GPU_MAX_SLOTS) + " device slots");
}

uint32_t ProgramBuilder::allocSlot() {
   return allocRun(1);
}

void ProgramBuilder::freeRun(uint32_t slot, uint32_t count) {
   const uint32_t mask = (count == 32 ? 0xFFFFFFFFu : ((1u << count) - 1u)) << slot;
   used_slots &= ~mask;
}

void ProgramBuilder::freeSlot(uint32_t slot) {
   if (!isLeafOperand(slot)) {  // leaf operands are not allocated
      freeRun(slot, 1);
   }
}

void ProgramBuilder::emit(uint32_t op, uint32_t dst, uint32_t a, uint32_t b, uint32_t imm) {
   code.push_back(op | (dst << 8) | (a << 16) | (b << 24));
   code.push_back(imm);
}

uint32_t ProgramBuilder::leaf(const uint64_t* device_bitset) {
   const auto found = std::find(leaves.begin(), leaves.end(), device_bitset);
   if (found != leaves.end()) {
      return static_cast<uint32_t>(found - leaves.begin());
   }
   leaves.push_back(device_bitset);
   return static_cast<uint32_t>(leaves.size() - 1);
}

uint32_t ProgramBuilder::sparseLeaf(uint32_t seqstore_id, uint32_t position, uint32_t symbol) {
   return leaf(sparsePointer(seqstore_id, position, symbol));
}

uint32_t ProgramBuilder::leafRun(const std::vector<const uint64_t*>& columns) {
   if (leaves.size() + columns.size() > SILO_GPU_MAX_LEAVES) {
      throw QueryCompilationException(
         "Compilation Error: filter expression reads more than " + std::to_string(SILO_GPU_MAX_LEAVES) + " stored columns in one device program"
      );
   }
   const auto first = static_cast<uint32_t>(leaves.size());
   leaves.insert(leaves.end(), columns.begin(), columns.end());  // consecutive, not de-duplicated
   return first | (static_cast<uint32_t>(columns.size()) << 16);
}

ProgramBuilder::Split ProgramBuilder::split(const operators::OperatorVector& children) {
   Split out;
   for (const auto& child : children) {
      if (const uint64_t* column = child->storedColumn(*this); column != nullptr) {
         out.columns.push_back(column);
      } else {
         out.composite.push_back(child.get());
      }
   }
   return out;
}

const uint64_t* ProgramBuilder::sparsePointer(uint32_t seqstore_id, uint32_t position, uint32_t symbol) {
   // Sparse planes are immutable once the store is finalised, so a materialised plane is kept for later
   // queries (bounded: beyond the budget a plane is built into a pooled temporary for this launch only).
   const DatabasePartition& partition = *rows.partition;
   const size_t row_bytes = static_cast<size_t>(partition.rowWords()) * sizeof(uint64_t);
   const uint64_t key = (static_cast<uint64_t>(seqstore_id) << 40) | (static_cast<uint64_t>(position) << 8) | symbol;
   {
      const std::shared_lock<std::shared_mutex> lock(partition.sparse_cache_mutex);
      const auto found = partition.sparse_cache.find(key);
      if (found != partition.sparse_cache.end()) {
         return found->second.as<uint64_t>();
      }
   }
   DeviceBuffer buffer = partition.pool.acquire(row_bytes);
   checkGpu(
      silo_gpu_store_sparse_plane(partition.store, seqstore_id, position, symbol, buffer.as<uint64_t>(), queryStream()),
      "silo_gpu_store_sparse_plane"
   );
   const uint64_t* pointer = buffer.as<uint64_t>();
   // other threads (on their own streams) may pick the plane up from the cache at once: finish it first
   checkGpu(silo_gpu_stream_synchronize(queryStream()), "silo_gpu_stream_synchronize");
   {
      const std::unique_lock<std::shared_mutex> lock(partition.sparse_cache_mutex);
      if ((partition.sparse_cache.size() + 1) * row_bytes <= DatabasePartition::SPARSE_CACHE_BYTES &&
          partition.sparse_cache.find(key) == partition.sparse_cache.end()) {
         partition.sparse_cache.emplace(key, std::move(buffer));
         return pointer;
      }
   }
   temporaries.push_back(std::move(buffer));
   return pointer;
}

uint64_t* ProgramBuilder::temporaryBitset() {
   const DatabasePartition& partition = *rows.partition;
   DeviceBuffer buffer = partition.pool.acquire(static_cast<size_t>(partition.rowWords()) * sizeof(uint64_t));
   auto* pointer = buffer.as<uint64_t>();
   temporaries.push_back(std::move(buffer));
   return pointer;
}

uint32_t ProgramBuilder::lowerChild(const operators::Operator& child) {
   const operators::Cost cost = child.cost();
   const size_t instructions_after = code.size() / 2 + cost.instructions + 8;
   const size_t leaves_after = leaves.size() + cost.leaves + 1;
   const bool fits = instructions_after <= SILO_GPU_MAX_INSTRUCTIONS && leaves_after <= SILO_GPU_MAX_LEAVES;
   const bool child_alone_fits = cost.instructions + 8 <= SILO_GPU_MAX_INSTRUCTIONS && cost.leaves + 1 <= SILO_GPU_MAX_LEAVES;
   if (fits || !child_alone_fits) {
      // too big even alone: descend — its own children are materialised one level down
      return child.lower(*this);
   }
   OperatorResult result = child.evaluate();
   const uint32_t operand = SILO_GPU_LEAF_OPERAND + leaf(result.bitset());
   materialized_children.push_back(std::move(result));
   return operand;
}

silo_gpu_bitprog ProgramBuilder::finishProgram(uint32_t result_slot) {
   if (result_slot != 0) {
      emit(SILO_GPU_OP_MOV, 0, result_slot);
      high_water = std::max(high_water, 1u);
   }
   if (code.size() / 2 > SILO_GPU_MAX_INSTRUCTIONS || leaves.size() > SILO_GPU_MAX_LEAVES) {
      throw QueryCompilationException("Compilation Error: filter expression does not fit one device program");
   }
   silo_gpu_bitprog program{};
   program.n_instructions = static_cast<uint32_t>(code.size() / 2);
   program.code = code.data();
   program.n_leaves = static_cast<uint32_t>(leaves.size());
   program.leaves = leaves.data();
   program.n_slots = std::max(high_water, 1u);
   return program;
}

void ProgramBuilder::run(uint32_t result_slot, uint64_t* out_bitset, uint64_t* out_count, void* stream) {
   const silo_gpu_bitprog program = finishProgram(result_slot);
   checkGpu(silo_gpu_filter_eval(rows.partition->store, &program, out_bitset, out_count, stream), "silo_gpu_filter_eval");
   if (!temporaries.empty() || !materialized_children.empty()) {
      // temporaries go back to the pool when the builder dies: make sure the kernel is done with them
      checkGpu(silo_gpu_stream_synchronize(stream), "silo_gpu_stream_synchronize");
   }
}

namespace {
/// One count slot per host thread (a slot serves one launch at a time).  Never destroyed: thread exit may come
/// after the HIP runtime has shut down.
silo_gpu_count_slot* threadCountSlot() {
   thread_local silo_gpu_count_slot* slot = nullptr;
   if (slot == nullptr) {
      checkGpu(silo_gpu_count_slot_create(&slot), "silo_gpu_count_slot_create");
   }
   return slot;
}
}  // namespace

uint32_t ProgramBuilder::runCounting(uint32_t result_slot, uint64_t* out_bitset, void* stream) {
   const silo_gpu_bitprog program = finishProgram(result_slot);
   silo_gpu_count_slot* slot = threadCountSlot();
   checkGpu(silo_gpu_filter_eval_count(rows.partition->store, &program, out_bitset, slot, stream), "silo_gpu_filter_eval_count");
   uint64_t count = 0;
   // the total arrives when the last block is done: every block has read its leaves by then, so the temporaries of
   // this builder may go back to the pool without a stream synchronisation
   checkGpu(silo_gpu_count_slot_wait(slot, &count, stream), "silo_gpu_count_slot_wait");
   return static_cast<uint32_t>(count);
}

namespace operators {

OperatorResult Operator::evaluate() const {
   return OperatorResult(rows, copy());
}

OperatorResult Operator::evaluate(std::unique_ptr<Operator> root) {
   const RowSpace rows = root->rows;
   return OperatorResult(rows, std::move(root));
}

namespace {

OperatorVector copyAll(const OperatorVector& source) {
   OperatorVector out;
   out.reserve(source.size());
   for (const auto& child : source) {
      out.push_back(child->copy());
   }
   return out;
}

Cost sumCost(const OperatorVector& a, const OperatorVector* b = nullptr) {
   Cost total;
   for (const auto& child : a) {
      const Cost c = child->cost();
      total.instructions += c.instructions + 1;
      total.leaves += c.leaves;
   }
   if (b != nullptr) {
      for (const auto& child : *b) {
         const Cost c = child->cost();
         total.instructions += c.instructions + 2;
         total.leaves += c.leaves;
      }
   }
   return total;
}

}  // namespace

// ---- Empty / Full (empty.cpp, full.cpp:24-28) ---------------------------------------------------
std::unique_ptr<Operator> Empty::copy() const {
   return std::make_unique<Empty>(rows);
}
std::unique_ptr<Operator> Empty::negate() const {
   return std::make_unique<Full>(rows);
}
uint32_t Empty::lower(ProgramBuilder& builder) const {
   const uint32_t slot = builder.allocSlot();
   builder.emit(SILO_GPU_OP_ZERO, slot);
   return slot;
}

std::unique_ptr<Operator> Full::copy() const {
   return std::make_unique<Full>(rows);
}
std::unique_ptr<Operator> Full::negate() const {
   return std::make_unique<Empty>(rows);
}
uint32_t Full::lower(ProgramBuilder& builder) const {
   const uint32_t slot = builder.allocSlot();
   builder.emit(SILO_GPU_OP_ONES, slot);
   return slot;
}

// ---- IndexScan (index_scan.cpp:28-30) -----------------------------------------------------------
std::unique_ptr<Operator> IndexScan::copy() const {
   if (sparse) {
      return std::make_unique<IndexScan>(seqstore_id, position, symbol, rows);
   }
   auto copied = std::make_unique<IndexScan>(bitmap, rows);
   copied->received = received;
   return copied;
}
std::unique_ptr<Operator> IndexScan::negate() const {
   return std::make_unique<Complement>(copy(), rows);
}
const uint64_t* IndexScan::storedColumn(ProgramBuilder& builder) const {
   return sparse ? builder.sparsePointer(seqstore_id, position, symbol) : bitmap;
}
uint32_t IndexScan::lower(ProgramBuilder& builder) const {
   // a leaf operand: the kernel stages every leaf in LDS up front, no instruction is emitted here
   const uint32_t index = sparse ? builder.sparseLeaf(seqstore_id, position, symbol) : builder.leaf(bitmap);
   return SILO_GPU_LEAF_OPERAND + index;
}

// ---- Selection (selection.cpp) ---------------------------------------------------------------------
Predicate Predicate::negated() const {  // selection.cpp:195-220: NOT (x >= v) becomes x < v — not the same for a NaN row
   static constexpr int NEGATED[] = {
      SILO_GPU_CMP_NOT_EQUALS, SILO_GPU_CMP_EQUALS, SILO_GPU_CMP_HIGHER_OR_EQUALS, SILO_GPU_CMP_LESS, SILO_GPU_CMP_LESS_OR_EQUALS,
      SILO_GPU_CMP_HIGHER};
   Predicate out = *this;
   out.comparator = NEGATED[comparator];
   return out;
}

std::unique_ptr<Operator> Selection::copy() const {
   return std::make_unique<Selection>(child != nullptr ? child->copy() : nullptr, predicates, rows);
}

std::unique_ptr<Operator> Selection::negate() const {  // selection.cpp:125-130
   if (child == nullptr && predicates.size() == 1) {
      return std::make_unique<Selection>(std::vector<Predicate>{predicates.at(0).negated()}, rows);
   }
   return std::make_unique<Complement>(this->copy(), rows);
}

Cost Selection::cost() const {
   Cost total{static_cast<uint32_t>(predicates.size()) + 1, static_cast<uint32_t>(predicates.size())};
   if (child != nullptr) {
      const Cost child_cost = child->cost();
      total.instructions += child_cost.instructions;
      total.leaves += child_cost.leaves;
   }
   return total;
}

uint32_t Selection::lower(ProgramBuilder& builder) const {
   // The reference probes every predicate row by row (selection.cpp:88-108).  Here each predicate is one pass of
   // k_bitset_from_compare over its column (4-8 bytes per row, on the query's stream, ahead of the fused program),
   // and the program ANDs the resulting bitsets like any other stored columns.
   std::vector<const uint64_t*> columns;
   for (const Predicate& predicate : predicates) {
      uint64_t* bitset = builder.temporaryBitset();
      checkGpu(
         silo_gpu_bitset_from_compare(
            rows.partition->store, bitset, predicate.column->deviceValues(), predicate.column->deviceValueType(), predicate.comparator,
            &predicate.value, queryStream()
         ),
         "silo_gpu_bitset_from_compare"
      );
      columns.push_back(bitset);
   }
   uint32_t acc = 0;
   bool have = false;
   if (columns.size() >= 2) {
      acc = builder.allocSlot();
      builder.emit(SILO_GPU_OP_AND_N, acc, 0, 0, builder.leafRun(columns));
      have = true;
   } else if (columns.size() == 1) {
      acc = SILO_GPU_LEAF_OPERAND + builder.leaf(columns[0]);
      have = true;
   }
   if (child != nullptr) {
      const uint32_t child_operand = builder.lowerChild(*child);
      if (!have) {
         return child_operand;
      }
      const uint32_t dst = isLeafOperand(acc) ? (isLeafOperand(child_operand) ? builder.allocSlot() : child_operand) : acc;
      builder.emit(SILO_GPU_OP_AND, dst, acc, child_operand);
      if (dst != child_operand && !isLeafOperand(child_operand)) {
         builder.freeSlot(child_operand);
      }
      return dst;
   }
   if (!have) {  // no predicate at all: every row
      const uint32_t slot = builder.allocSlot();
      builder.emit(SILO_GPU_OP_ONES, slot);
      return slot;
   }
   return acc;
}

// ---- BitmapProducer (bitmap_producer.cpp) -----------------------------------------------------------
std::unique_ptr<Operator> BitmapProducer::copy() const {
   return std::make_unique<BitmapProducer>(index, membership, rows);
}
std::unique_ptr<Operator> BitmapProducer::negate() const {
   return std::make_unique<Complement>(this->copy(), rows);
}
uint32_t BitmapProducer::lower(ProgramBuilder& builder) const {
   uint64_t* bitset = builder.temporaryBitset();
   checkGpu(
      silo_gpu_bitset_from_pairs(
         rows.partition->store, bitset, index->device_rows, index->device_ids, static_cast<uint32_t>(index->pair_rows.size()),
         membership.data(), static_cast<uint32_t>(membership.size()), queryStream()
      ),
      "silo_gpu_bitset_from_pairs"
   );
   return SILO_GPU_LEAF_OPERAND + builder.leaf(bitset);
}

// ---- BitmapSelection (bitmap_selection.cpp:33-71) -----------------------------------------------
std::unique_ptr<Operator> BitmapSelection::copy() const {
   auto out = std::make_unique<BitmapSelection>(*this);
   return out;
}
std::unique_ptr<Operator> BitmapSelection::negate() const {
   auto out = std::make_unique<BitmapSelection>(*this);
   out->comparator = comparator == CONTAINS ? NOT_CONTAINS : CONTAINS;
   return out;
}
const uint64_t* BitmapSelection::storedColumn(ProgramBuilder& builder) const {
   if (comparator != CONTAINS) {
      return nullptr;
   }
   return materialise ? builder.sparsePointer(seqstore_id, local_position, symbol) : missing_plane;
}
uint32_t BitmapSelection::lower(ProgramBuilder& builder) const {
   const uint32_t operand = SILO_GPU_LEAF_OPERAND + (materialise ? builder.sparseLeaf(seqstore_id, local_position, symbol) : builder.leaf(missing_plane));
   if (comparator == CONTAINS) {
      return operand;
   }
   const uint32_t slot = builder.allocSlot();
   builder.emit(SILO_GPU_OP_NOT, slot, operand);
   return slot;
}

// ---- Complement (complement.cpp) ----------------------------------------------------------------
std::unique_ptr<Operator> Complement::copy() const {
   return std::make_unique<Complement>(child->copy(), rows);
}
std::unique_ptr<Operator> Complement::negate() const {
   return child->copy();
}
uint32_t Complement::lower(ProgramBuilder& builder) const {  // flip(0,row_count): complement.cpp:50-54
   const uint32_t operand = builder.lowerChild(*child);
   const uint32_t slot = isLeafOperand(operand) ? builder.allocSlot() : operand;
   builder.emit(SILO_GPU_OP_NOT, slot, operand);
   return slot;
}
Cost Complement::cost() const {
   Cost c = child->cost();
   c.instructions += 1;
   return c;
}

// ---- Intersection (intersection.cpp) ------------------------------------------------------------
Intersection::Intersection(OperatorVector&& children_, OperatorVector&& negated_children_, RowSpace rows)
    : Operator(rows), children(std::move(children_)), negated_children(std::move(negated_children_)) {
   if (children.empty()) {
      throw QueryCompilationException(
         "Compilation bug: Intersection without non-negated children is not allowed. Should be compiled as a union."
      );
   }
   if (children.size() + negated_children.size() < 2) {
      throw QueryCompilationException("Compilation bug: Intersection needs at least two children.");
   }
}
std::string Intersection::toString() const {
   std::string res = "(" + children[0]->toString();
   for (size_t i = 1; i < children.size(); ++i) {
      res += " & " + children[i]->toString();
   }
   for (const auto& child : negated_children) {
      res += " &! " + child->toString();
   }
   return res + ")";
}
std::unique_ptr<Operator> Intersection::copy() const {
   return std::make_unique<Intersection>(copyAll(children), copyAll(negated_children), rows);
}
std::unique_ptr<Operator> Intersection::negate() const {
   return std::make_unique<Complement>(copy(), rows);
}
uint32_t Intersection::lower(ProgramBuilder& builder) const {
   // The reference orders children by cardinality to keep roaring intermediates small
   // (intersection.cpp:94-108); a dense word-parallel AND has no such sensitivity.  Children that are
   // stored columns are folded by ONE n-ary instruction (streamed 8 loads at a time).
   const ProgramBuilder::Split positive = builder.split(children);
   const ProgramBuilder::Split negative = builder.split(negated_children);
   constexpr uint32_t NONE = ~0u;
   uint32_t acc = NONE;
   const auto combine = [&](uint32_t op, uint32_t tmp) {
      if (acc == NONE) {
         acc = tmp;
         return;
      }
      const uint32_t dst = isLeafOperand(acc) ? (isLeafOperand(tmp) ? builder.allocSlot() : tmp) : acc;
      builder.emit(op, dst, acc, tmp);
      if (dst != tmp) {
         builder.freeSlot(tmp);
      }
      acc = dst;
   };
   if (positive.columns.size() >= 2) {
      acc = builder.allocSlot();
      builder.emit(SILO_GPU_OP_AND_N, acc, 0, 0, builder.leafRun(positive.columns));
   } else if (positive.columns.size() == 1) {
      acc = SILO_GPU_LEAF_OPERAND + builder.leaf(positive.columns[0]);
   }
   for (const Operator* child : positive.composite) {
      combine(SILO_GPU_OP_AND, builder.lowerChild(*child));
   }
   // children is never empty (constructor), so acc is set here
   if (negative.columns.size() >= 2) {
      const uint32_t tmp = builder.allocSlot();
      builder.emit(SILO_GPU_OP_OR_N, tmp, 0, 0, builder.leafRun(negative.columns));
      combine(SILO_GPU_OP_ANDNOT, tmp);
   } else if (negative.columns.size() == 1) {
      combine(SILO_GPU_OP_ANDNOT, SILO_GPU_LEAF_OPERAND + builder.leaf(negative.columns[0]));
   }
   for (const Operator* child : negative.composite) {
      combine(SILO_GPU_OP_ANDNOT, builder.lowerChild(*child));
   }
   return acc;
}
Cost Intersection::cost() const {
   return sumCost(children, &negated_children);
}

// ---- Union (union.cpp) --------------------------------------------------------------------------
std::string Union::toString() const {
   if (children.empty()) {
      return "()";
   }
   std::string res = "(" + children[0]->toString();
   for (size_t i = 1; i < children.size(); ++i) {
      res += " | " + children[i]->toString();
   }
   return res + ")";
}
std::unique_ptr<Operator> Union::copy() const {
   return std::make_unique<Union>(copyAll(children), rows);
}
std::unique_ptr<Operator> Union::negate() const {
   return std::make_unique<Complement>(copy(), rows);
}
uint32_t Union::lower(ProgramBuilder& builder) const {
   if (children.empty()) {  // union.test.cpp:28-34: the union of nothing is empty
      const uint32_t slot = builder.allocSlot();
      builder.emit(SILO_GPU_OP_ZERO, slot);
      return slot;
   }
   const ProgramBuilder::Split parts = builder.split(children);
   constexpr uint32_t NONE = ~0u;
   uint32_t acc = NONE;
   if (parts.columns.size() >= 2) {  // roaring fastunion (union.cpp:44) -> one streamed n-ary OR
      acc = builder.allocSlot();
      builder.emit(SILO_GPU_OP_OR_N, acc, 0, 0, builder.leafRun(parts.columns));
   } else if (parts.columns.size() == 1) {
      acc = SILO_GPU_LEAF_OPERAND + builder.leaf(parts.columns[0]);
   }
   for (const Operator* child : parts.composite) {
      const uint32_t tmp = builder.lowerChild(*child);
      if (acc == NONE) {
         acc = tmp;
         continue;
      }
      const uint32_t dst = isLeafOperand(acc) ? (isLeafOperand(tmp) ? builder.allocSlot() : tmp) : acc;
      builder.emit(SILO_GPU_OP_OR, dst, acc, tmp);
      if (dst != tmp) {
         builder.freeSlot(tmp);
      }
      acc = dst;
   }
   return acc;
}
Cost Union::cost() const {
   Cost c = sumCost(children);
   c.instructions += 1;
   return c;
}

// ---- Threshold (threshold.cpp) ------------------------------------------------------------------
Threshold::Threshold(OperatorVector&& non_negated_children_, OperatorVector&& negated_children_, uint32_t number_of_matchers, bool match_exactly, RowSpace rows)
    : Operator(rows),
      non_negated_children(std::move(non_negated_children_)),
      negated_children(std::move(negated_children_)),
      number_of_matchers(number_of_matchers),
      match_exactly(match_exactly) {
   if (number_of_matchers >= non_negated_children.size() + negated_children.size()) {  // threshold.cpp:28-33
      throw QueryCompilationException(
         "Compilation Error: number_of_matchers must be less than the number of children of a threshold expression"
      );
   }
   if (number_of_matchers == 0) {
      throw QueryCompilationException("Compilation Error: number_of_matchers must be greater than zero");
   }
}
std::string Threshold::toString() const {
   std::string res = match_exactly ? "=" : ">=";
   for (const auto& child : non_negated_children) {
      res += ", " + child->toString();
   }
   for (const auto& child : negated_children) {
      res += ", ! " + child->toString();
   }
   return res + ")";
}
std::unique_ptr<Operator> Threshold::copy() const {
   return std::make_unique<Threshold>(copyAll(non_negated_children), copyAll(negated_children), number_of_matchers, match_exactly, rows);
}
std::unique_ptr<Operator> Threshold::negate() const {
   return std::make_unique<Complement>(copy(), rows);
}
uint32_t Threshold::lower(ProgramBuilder& builder) const {
   // The reference keeps a DP table of n (or n+1) bitmaps, table[j] = rows matched by > j children
   // so far (threshold.cpp:64-138).  Word-parallel restatement: a bit-sliced per-row counter of
   // ceil(log2(k+1)) slots, one ripple-carry add per child, one comparison with n at the end.
   const uint32_t k = static_cast<uint32_t>(non_negated_children.size() + negated_children.size());
   const uint32_t bits = static_cast<uint32_t>(std::bit_width(k));
   const uint32_t counter = builder.allocRun(bits);
   for (uint32_t bit = 0; bit < bits; ++bit) {
      builder.emit(SILO_GPU_OP_ZERO, counter + bit);
   }
   const ProgramBuilder::Split positive = builder.split(non_negated_children);
   const ProgramBuilder::Split negative = builder.split(negated_children);
   if (!positive.columns.empty()) {
      builder.emit(SILO_GPU_OP_CNT_ADD_N, counter, 0, bits, builder.leafRun(positive.columns));
   }
   if (!negative.columns.empty()) {
      builder.emit(SILO_GPU_OP_CNT_ADD_NOT_N, counter, 0, bits, builder.leafRun(negative.columns));
   }
   for (const Operator* child : positive.composite) {
      const uint32_t tmp = builder.lowerChild(*child);
      builder.emit(SILO_GPU_OP_CNT_ADD, counter, tmp, bits);
      builder.freeSlot(tmp);
   }
   for (const Operator* child : negative.composite) {
      const uint32_t operand = builder.lowerChild(*child);
      const uint32_t tmp = isLeafOperand(operand) ? builder.allocSlot() : operand;
      builder.emit(SILO_GPU_OP_NOT, tmp, operand);
      builder.emit(SILO_GPU_OP_CNT_ADD, counter, tmp, bits);
      builder.freeSlot(tmp);
   }
   const uint32_t result = builder.allocSlot();
   builder.emit(match_exactly ? SILO_GPU_OP_CNT_EQ : SILO_GPU_OP_CNT_GE, result, counter, bits, number_of_matchers);
   builder.freeRun(counter, bits);
   return result;
}
Cost Threshold::cost() const {
   Cost c = sumCost(non_negated_children, &negated_children);
   c.instructions += 8;
   return c;
}

}  // namespace operators
}  // namespace silo::query_engine
