// actions.cpp — Action base (ordering / limit / offset), Aggregated (count) and Mutations<SymbolType>.
// Reference: src/silo/query_engine/actions/{action,aggregated,mutations}.cpp.
#include <algorithm>
#include <charconv>
#include <cmath>
#include <functional>
#include <string_view>
#include <tuple>

#include "query_engine.h"

namespace silo::query_engine::actions {

// ---- Action (action.cpp:37-117) ------------------------------------------------------------------
void Action::applySort(QueryResult& result) const {  // action.cpp:37-66
   if (order_by_fields.empty()) {
      return;
   }
   // The reference's comparator looks every field up in the row's std::map on every comparison; here the fields are
   // looked up once per row and the sort runs over (row index, field pointers).  Same ordering, and like the reference
   // only the rows the offset / limit window reaches are put in order (a partial sort).
   using Field = std::optional<std::variant<std::string, int32_t, double>>;
   std::vector<QueryResultEntry>& rows = result.query_result;
   const size_t n_keys = order_by_fields.size();
   std::vector<const Field*> keys(rows.size() * n_keys);
   std::vector<uint32_t> order(rows.size());
   for (size_t row = 0; row < rows.size(); ++row) {
      order[row] = static_cast<uint32_t>(row);
      for (size_t k = 0; k < n_keys; ++k) {
         keys[row * n_keys + k] = &rows[row].fields.at(order_by_fields[k].name);
      }
   }
   const auto before = [&](uint32_t left, uint32_t right) {
      for (size_t k = 0; k < n_keys; ++k) {
         const Field& a = *keys[left * n_keys + k];
         const Field& b = *keys[right * n_keys + k];
         if (!(a == b)) {
            return (a < b) == order_by_fields[k].ascending;
         }
      }
      return false;
   };
   const size_t window_end = limit.has_value() ? std::min<size_t>(rows.size(), static_cast<size_t>(*limit) + offset.value_or(0)) : rows.size();
   if (window_end < rows.size()) {
      std::partial_sort(order.begin(), order.begin() + static_cast<std::ptrdiff_t>(window_end), order.end(), before);
   } else {
      std::sort(order.begin(), order.end(), before);
   }
   std::vector<QueryResultEntry> sorted;
   sorted.reserve(rows.size());
   for (const uint32_t row : order) {
      sorted.push_back(std::move(rows[row]));
   }
   rows = std::move(sorted);
}

void Action::applyOffsetAndLimit(QueryResult& result) const {
   // the window [offset, offset + limit) of the (sorted) rows, clipped to what there is: action.cpp:68-95
   auto& rows = result.query_result;
   const size_t first = std::min<size_t>(offset.value_or(0), rows.size());
   const size_t count = limit.has_value() ? std::min<size_t>(*limit, rows.size() - first) : rows.size() - first;
   if (first != 0) {
      std::move(rows.begin() + static_cast<std::ptrdiff_t>(first), rows.begin() + static_cast<std::ptrdiff_t>(first + count), rows.begin());
   }
   rows.resize(count);
}

void Action::setOrdering(const std::vector<OrderByField>& fields, std::optional<uint32_t> row_limit, std::optional<uint32_t> row_offset) {
   order_by_fields = fields;
   limit = row_limit;
   offset = row_offset;
}

QueryResult Action::orderAndLimit(QueryResult result) const {  // action.cpp:110-116
   if (offset.value_or(0) < result.query_result.size()) {
      applySort(result);  // sorts no further than the window needs
   }
   applyOffsetAndLimit(result);  // an offset past the end leaves no row
   return result;
}

QueryResult Action::executeAndOrder(const Database& database, std::vector<OperatorResult> bitmap_filter) const {
   validateOrderByFields(database);
   return orderAndLimit(execute(database, std::move(bitmap_filter)));
}

std::unique_ptr<Action::Pending> Action::begin(const Database& database, std::vector<OperatorResult> bitmap_filter) const {
   validateOrderByFields(database);
   auto pending = std::make_unique<Pending>();
   pending->bitmap_filter = std::move(bitmap_filter);
   return pending;
}

QueryResult Action::finish(const Database& database, Pending& pending) const {
   return orderAndLimit(execute(database, std::move(pending.bitmap_filter)));
}

std::string Action::finishJson(const Database& database, Pending& pending) const {
   return toJsonText(finish(database, pending));
}

// ---- ScanBatcher ---------------------------------------------------------------------------------
namespace {

/// Sums `n` uint32 across ranks in place on the device (no-op for a single rank).
void allReduce(const Database& database, uint32_t* device_values, size_t n) {
   if (database.all_reduce != nullptr) {  // also with a single rank: lets a 1-GPU box exercise the collective path
      const int status = database.all_reduce(database.all_reduce_context, device_values, n, queryStream());
      if (status != 0) {
         throw DeviceException("all-reduce of counts failed with status " + std::to_string(status) + (status < 0 ? std::string(": ") + silo_gpu_last_error() : ""));
      }
   }
}

/// Words an all-reduce of a query carries behind its payload: the two halves of the query's fingerprint.
constexpr size_t FINGERPRINT_WORDS = 2;

/// Writes the calling thread's query fingerprint behind `n_payload` words of an all-reduce buffer (stream-ordered).
void appendFingerprint(uint32_t* device_values, size_t n_payload, uint64_t fingerprint) {
   const uint32_t halves[FINGERPRINT_WORDS] = {static_cast<uint32_t>(fingerprint), static_cast<uint32_t>(fingerprint >> 32)};
   checkGpu(silo_gpu_memcpy_h2d(device_values + n_payload, halves, sizeof(halves), queryStream()), "silo_gpu_memcpy_h2d");
}

/// After the all-reduce: every rank added the same fingerprint, or the ranks ran different queries (their collectives
/// paired up all the same and the payload is a mix) — refuse the result.
void checkSameQuery(const Database& database, const uint32_t* summed_halves, uint64_t fingerprint) {
   const uint32_t world = std::max<uint32_t>(1, database.shard_world);
   const bool same = summed_halves[0] == static_cast<uint32_t>(fingerprint) * world &&
                     summed_halves[1] == static_cast<uint32_t>(fingerprint >> 32) * world;
   if (!same) {
      throw DeviceException(
         "the ranks of this sharded database did not run the same query: every rank has to run the same queries in the same order "
         "(query fingerprints differ in the all-reduce)"
      );
   }
}

}  // namespace

namespace {
thread_local ScanBatcher* g_active_batcher = nullptr;
}

ScanBatcher::ScanBatcher() : previous(g_active_batcher) {
   g_active_batcher = this;
}

ScanBatcher::~ScanBatcher() {
   g_active_batcher = previous;
}

ScanBatcher* ScanBatcher::active() {
   return g_active_batcher;
}

void ScanBatcher::flush() {
   // Per (store, filter) the position ranges it is to be scanned over, in recording order; filters with the same list
   // of ranges — the queries of a batch that ask for the same sequence stores — share one multi-range call, in which
   // the planes of every range are streamed once for all of them (K1c) and every filter is looked at once (K1s).
   struct FilterScans {
      silo_gpu_store* store;
      const uint64_t* filter;
      std::vector<silo_gpu_scan_range> ranges;
      std::vector<uint32_t*> counts;
   };
   const auto same_range = [](const silo_gpu_scan_range& a, const silo_gpu_scan_range& b) {
      return a.seqstore_id == b.seqstore_id && a.pos_begin == b.pos_begin && a.pos_end == b.pos_end;
   };
   std::vector<FilterScans> by_filter;
   for (const Request& request : requests) {
      if (request.filter == nullptr) {  // full filter: cached totals, no pass over the planes
         checkGpu(
            silo_gpu_mutations_scan(request.store, request.seqstore_id, nullptr, request.pos_begin, request.pos_end, request.counts, queryStream()),
            "silo_gpu_mutations_scan"
         );
         continue;
      }
      const silo_gpu_scan_range range{request.seqstore_id, request.pos_begin, request.pos_end};
      FilterScans* entry = nullptr;
      for (FilterScans& candidate : by_filter) {
         if (candidate.store == request.store && candidate.filter == request.filter &&
             std::none_of(candidate.ranges.begin(), candidate.ranges.end(), [&](const auto& other) { return same_range(other, range); })) {
            entry = &candidate;
            break;
         }
      }
      if (entry == nullptr) {
         by_filter.push_back({request.store, request.filter, {}, {}});
         entry = &by_filter.back();
      }
      entry->ranges.push_back(range);
      entry->counts.push_back(request.counts);
   }
   std::vector<bool> done(by_filter.size(), false);
   for (size_t i = 0; i < by_filter.size(); ++i) {
      if (done[i]) {
         continue;
      }
      const FilterScans& first = by_filter[i];
      std::vector<size_t> group;
      for (size_t k = i; k < by_filter.size(); ++k) {
         const FilterScans& other = by_filter[k];
         if (!done[k] && other.store == first.store && other.ranges.size() == first.ranges.size() &&
             std::equal(other.ranges.begin(), other.ranges.end(), first.ranges.begin(), same_range)) {
            done[k] = true;
            group.push_back(k);
         }
      }
      std::vector<const uint64_t*> filters;
      for (const size_t k : group) {
         filters.push_back(by_filter[k].filter);
      }
      std::vector<uint32_t*> counts;  // [range][filter]
      for (size_t r = 0; r < first.ranges.size(); ++r) {
         for (const size_t k : group) {
            counts.push_back(by_filter[k].counts[r]);
         }
      }
      checkGpu(
         silo_gpu_mutations_scan_ranges(
            first.store, first.ranges.data(), static_cast<uint32_t>(first.ranges.size()), filters.data(), static_cast<uint32_t>(filters.size()),
            counts.data(), queryStream()
         ),
         "silo_gpu_mutations_scan_ranges"
      );
   }
   requests.clear();
   for (const Reduction& reduction : reductions) {
      allReduce(*reduction.database, reduction.device_values, reduction.n);
   }
   reductions.clear();
   for (const auto& callback : after_flush) {
      callback();
   }
   after_flush.clear();
}

// ---- Aggregated (aggregated.cpp:58-96; group-by is outside the hot path) -----------------------------
void Aggregated::validateOrderByFields(const Database& database) const {  // aggregated.cpp:26-38,71-88
   for (const std::string& group_by_field : group_by_fields) {
      CHECK_SILO_QUERY(
         database.database_config.getMetadata(group_by_field).has_value(), "Metadata field '" + group_by_field + "' to group by not found"
      )
   }
   for (const OrderByField& field : order_by_fields) {
      CHECK_SILO_QUERY(
         field.name == "count" || std::find(group_by_fields.begin(), group_by_fields.end(), field.name) != group_by_fields.end(),
         "The orderByField '" + field.name + "' cannot be ordered by, as it does not appear in the groupByFields."
      )
   }
}

QueryResult Aggregated::execute(const Database& database, std::vector<OperatorResult> bitmap_filters) const {
   if (!group_by_fields.empty()) {
      return aggregateWithGrouping(database, bitmap_filters);
   }
   uint32_t count = 0;  // aggregateWithoutGrouping, aggregated.cpp:58-66
   for (const auto& filter : bitmap_filters) {
      count += filter.cardinality();
   }
   if (database.shard_world > 1 && !database.shard_by_position && database.all_reduce != nullptr && !database.partitions.empty()) {
      // sequence-id sharding: every rank holds different rows
      const DatabasePartition& partition = database.partitions.front();
      const uint64_t fingerprint = Database::queryFingerprint();
      uint32_t words[1 + FINGERPRINT_WORDS] = {count, static_cast<uint32_t>(fingerprint), static_cast<uint32_t>(fingerprint >> 32)};
      DeviceBuffer buffer = partition.pool.acquire(sizeof(words));
      checkGpu(silo_gpu_memcpy_h2d(buffer.get(), words, sizeof(words), queryStream()), "silo_gpu_memcpy_h2d");
      allReduce(database, buffer.as<uint32_t>(), 1 + FINGERPRINT_WORDS);
      checkGpu(silo_gpu_memcpy_d2h(words, buffer.get(), sizeof(words), queryStream()), "silo_gpu_memcpy_d2h");
      checkSameQuery(database, words + 1, fingerprint);
      count = words[0];
   }
   std::map<std::string, std::optional<std::variant<std::string, int32_t, double>>> tuple_fields;
   tuple_fields["count"] = static_cast<int32_t>(count);
   return QueryResult{std::vector<QueryResultEntry>{{tuple_fields}}};
}

// ---- Mutations<SymbolType> (mutations.cpp) ---------------------------------------------------------
template <typename SymbolType>
std::map<std::string, typename Mutations<SymbolType>::PrefilteredBitmaps> Mutations<SymbolType>::preFilterBitmaps(
   const Database& database, std::vector<OperatorResult>& bitmap_filter
) {  // mutations.cpp:35-62: per sequence store the (filter, partition store) pairs to scan; empty filters drop out, filters
     // that select every row of their partition go on the list that is answered from stored totals
   std::map<std::string, PrefilteredBitmaps> per_store;
   size_t partition_index = 0;
   for (const DatabasePartition& partition : database.partitions) {
      OperatorResult& selected = bitmap_filter[partition_index++];
      selected.materialize();  // one launch yields both the bitset and its cardinality
      const uint32_t n_selected = selected.cardinality();
      if (n_selected == 0) {
         continue;
      }
      for (const auto& [name, store] : partition.getSequenceStores<SymbolType>()) {
         PrefilteredBitmaps& lists = per_store[name];
         (n_selected == partition.sequence_count ? lists.full_bitmaps : lists.bitmaps).emplace_back(selected, store);
      }
   }
   return per_store;
}

template <typename SymbolType>
void Mutations<SymbolType>::calculateMutationsPerPosition(
   const Database& database, const SequenceStore<SymbolType>& sequence_store, const PrefilteredBitmaps& bitmap_filter, uint32_t* device_counts
) {
   // mutations.cpp:139-164 runs and_cardinality(filter, column) per position x symbol under
   // tbb::parallel_for; here each (partition, sequence store) is ONE scan kernel (K1) that accumulates
   // into a single device table, exactly as the reference sums partitions into one table (:71,:108).
   const auto sequence_length = static_cast<uint32_t>(sequence_store.reference_sequence.size());
   constexpr uint32_t n_symbols = SymbolType::VALID_MUTATION_SYMBOLS.size();
   // position-range shard of this rank; [0, P) when not sharded by position
   const auto [pos_begin, pos_end] = database.positionWindow(sequence_length);
   uint32_t* window = device_counts + static_cast<size_t>(pos_begin) * n_symbols;
   // the device store of a rank holds exactly its window: local positions [0, pos_end - pos_begin)
   const uint32_t local_positions = pos_end - pos_begin;
   ScanBatcher* batcher = ScanBatcher::active();  // inside a batch of queries the scans are only recorded
   const auto scan = [&](const SequenceStorePartition<SymbolType>& store, const uint64_t* filter) {
      if (batcher != nullptr) {
         batcher->add({store.store, store.seqstore_id, filter, 0, local_positions, window});
      } else {
         checkGpu(
            silo_gpu_mutations_scan(store.store, store.seqstore_id, filter, 0, local_positions, window, queryStream()), "silo_gpu_mutations_scan"
         );
      }
   };
   for (const auto& [filter, store] : bitmap_filter.bitmaps) {
      scan(store, filter.bitset());
   }
   for (const auto& [filter, store] : bitmap_filter.full_bitmaps) {
      // full filter: the reference reads plain cardinalities (mutations.cpp:98-136); NULL = all rows
      scan(store, nullptr);
   }
}

template <typename SymbolType>
void Mutations<SymbolType>::validateOrderByFields(const Database& /*database*/) const {  // mutations.cpp:166-182
   // a Mutations row can be ordered by three of its four fields (not by sequenceName)
   for (const OrderByField& field : order_by_fields) {
      const bool sortable = field.name == MUTATION_FIELD_NAME || field.name == PROPORTION_FIELD_NAME || field.name == COUNT_FIELD_NAME;
      CHECK_SILO_QUERY(sortable, "OrderByField " + field.name + " is not contained in the result of this operation.")
   }
}

template <typename SymbolType>
void Mutations<SymbolType>::addMutationsToOutput(
   const std::string& sequence_name, const SequenceStore<SymbolType>& sequence_store, const uint32_t* counts,
   std::vector<QueryResultEntry>& output
) const {  // mutations.cpp:184-232, over the count table the device filled: counts[position][valid symbol]
   constexpr size_t n_symbols = SymbolType::VALID_MUTATION_SYMBOLS.size();
   const auto& reference = sequence_store.reference_sequence;
   for (size_t position = 0; position < reference.size(); ++position) {
      const uint32_t* cell = counts + position * n_symbols;
      uint32_t covered = 0;  // rows of the filter with a valid symbol here; uint32 wrap-around like the reference's sum
      for (size_t k = 0; k < n_symbols; ++k) {
         covered += cell[k];
      }
      if (covered == 0) {
         continue;
      }
      // a symbol is reported when count >= ceil(covered * minProportion); written as the reference writes it (IEEE double,
      // then "count > that - 1" in uint32), because the rounding decides rows at the boundary
      const uint32_t must_exceed = min_proportion == 0 ? 0 : static_cast<uint32_t>(std::ceil(static_cast<double>(covered) * min_proportion) - 1);
      for (size_t k = 0; k < n_symbols; ++k) {
         if (SymbolType::VALID_MUTATION_SYMBOLS[k] == reference[position] || cell[k] <= must_exceed) {
            continue;
         }
         addSelectedRowToOutput(
            sequence_name, sequence_store, static_cast<uint32_t>(position),
            silo_gpu_mutation_row{static_cast<uint32_t>(position), static_cast<uint32_t>(k), cell[k], covered}, output
         );
      }
   }
}

template <typename SymbolType>
void Mutations<SymbolType>::addSelectedRowToOutput(
   const std::string& sequence_name, const SequenceStore<SymbolType>& sequence_store, uint32_t position, const silo_gpu_mutation_row& row,
   std::vector<QueryResultEntry>& output
) const {  // the four fields of mutations.cpp:213-224 for a cell that k_mutations_select let through
   const char from = SymbolType::symbolToChar(sequence_store.reference_sequence.at(position));
   const char to = SymbolType::symbolToChar(SymbolType::VALID_MUTATION_SYMBOLS.at(row.symbol_index));
   QueryResultEntry& entry = output.emplace_back();
   // keys arrive in map order (count < mutation < proportion < sequenceName): every insert is at the end
   entry.fields.emplace_hint(entry.fields.end(), COUNT_FIELD_NAME, static_cast<int32_t>(row.count));
   entry.fields.emplace_hint(entry.fields.end(), MUTATION_FIELD_NAME, from + std::to_string(position + 1) + to);
   entry.fields.emplace_hint(entry.fields.end(), PROPORTION_FIELD_NAME, static_cast<double>(row.count) / static_cast<double>(row.total));
   entry.fields.emplace_hint(entry.fields.end(), SEQUENCE_FIELD_NAME, sequence_name);
}

template <typename SymbolType>
std::unique_ptr<Action::Pending> Mutations<SymbolType>::begin(const Database& database, std::vector<OperatorResult> bitmap_filter) const {
   // first half of mutations.cpp:234-272: validate, pre-filter, queue the scans of every requested store
   validateOrderByFields(database);
   auto pending = std::make_unique<PendingScans>();
   for (const auto& sequence_name : sequence_names) {
      CHECK_SILO_QUERY(
         database.getSequenceStores<SymbolType>().count(sequence_name) != 0,
         "Database does not contain the " + std::string(SymbolType::SYMBOL_NAME_LOWER_CASE) + " sequence with name: '" + sequence_name + "'"
      )
      pending->sequence_names.emplace_back(sequence_name);
   }
   if (sequence_names.empty()) {
      for (const auto& [sequence_name, _] : database.getSequenceStores<SymbolType>()) {
         pending->sequence_names.emplace_back(sequence_name);
      }
   }
   pending->bitmap_filter = std::move(bitmap_filter);  // the scans read these bitsets: they live as long as the scans
   std::map<std::string, PrefilteredBitmaps> bitmaps_to_evaluate = preFilterBitmaps(database, pending->bitmap_filter);
   Trace::mark("filter_materialized");

   // ONE count table for the query, all stores of the alphabet back to back (MutationTableLayout): one memset,
   // the scans of every requested store (in order on this thread's stream — or, inside a batch of queries, recorded
   // so that scans of different queries share passes over the planes), one all-reduce when sharded, one row
   // selection on the device and one transfer; finish() only waits for that transfer.
   const bool sharded = database.shard_world > 1 && database.all_reduce != nullptr;
   const MutationTableLayout& layout = database.getMutationTableLayout<SymbolType>();
   constexpr uint32_t n_symbols = SymbolType::VALID_MUTATION_SYMBOLS.size();
   const size_t n_counts = static_cast<size_t>(layout.total_positions) * n_symbols;
   if (database.partitions.empty() || n_counts == 0 || pending->sequence_names.empty() || (bitmaps_to_evaluate.empty() && !sharded)) {
      return pending;  // nothing selected (and no other rank to contribute): no rows
   }
   const bool reduced = database.all_reduce != nullptr;  // the table is all-reduced, with the query's fingerprint behind it
   pending->table_bytes = ((n_counts + (reduced ? FINGERPRINT_WORDS : 0)) * sizeof(uint32_t) + 15) / 16 * 16;
   pending->row_capacity = layout.reference_index_device != nullptr ? database.mutation_row_capacity : 0;
   pending->device_table = database.partitions.front().pool.acquire(pending->table_bytes);
   auto* device_counts = static_cast<uint32_t*>(pending->device_table.get());
   checkGpu(silo_gpu_memset_async(device_counts, 0, pending->table_bytes, queryStream()), "silo_gpu_memset_async");

   const PrefilteredBitmaps no_bitmaps{};
   std::vector<std::string> scanned;  // a store requested twice is scanned once
   // a query on its own still records its scans — one per sequence store — so that they leave as ONE multi-range call
   std::optional<ScanBatcher> own_batcher;
   if (ScanBatcher::active() == nullptr) {
      own_batcher.emplace();
   }
   try {
   for (const auto& sequence_name : pending->sequence_names) {
      if (std::find(scanned.begin(), scanned.end(), sequence_name) != scanned.end()) {
         continue;
      }
      scanned.push_back(sequence_name);
      const SequenceStore<SymbolType>& sequence_store = database.getSequenceStores<SymbolType>().at(sequence_name);
      const auto found = bitmaps_to_evaluate.find(sequence_name);
      calculateMutationsPerPosition(
         database, sequence_store, found != bitmaps_to_evaluate.end() ? found->second : no_bitmaps,
         device_counts + static_cast<size_t>(layout.position_offset.at(sequence_name)) * n_symbols
      );
   }
   Trace::mark("scan_launched");

   PendingScans& scans = *pending;
   const uint8_t* reference_index = layout.reference_index_device.get();
   const uint32_t total_positions = layout.total_positions;
   const double proportion = min_proportion;
   scans.fingerprint = Database::queryFingerprint();
   const auto select_and_fetch = [&scans, device_counts, reference_index, total_positions, proportion, reduced, n_counts]() {
      if (reduced) {
         scans.check_fetch = HostFetch(device_counts + n_counts, FINGERPRINT_WORDS * sizeof(uint32_t), queryStream());
      }
      if (scans.row_capacity == 0) {
         scans.fetch = HostFetch(device_counts, scans.table_bytes, queryStream());
         return;
      }
      // K4 picks the rows on the device (threshold arithmetic of mutations.cpp:197-211) and writes them straight into
      // page-locked host memory: only they travel, and nothing waits for a copy or an event
      scans.row_slot = RowSlot(scans.row_capacity);
      scans.row_slot.select(device_counts, reference_index, total_positions, n_symbols, proportion, queryStream());
   };
   ScanBatcher* batcher = ScanBatcher::active();
   if (reduced) {
      appendFingerprint(device_counts, n_counts, scans.fingerprint);
      batcher->addReduction(database, device_counts, n_counts + FINGERPRINT_WORDS);  // the whole query in one collective
   }
   batcher->afterFlush(select_and_fetch);  // the scans are only recorded so far
   if (own_batcher) {
      own_batcher->flush();
   }
   } catch (...) {
      // launches of this query may be in flight on the stream: let them finish before its buffers return to the pool
      (void)silo_gpu_stream_synchronize(queryStream());
      throw;
   }
   return pending;
}

template <typename SymbolType>
std::vector<typename Mutations<SymbolType>::SelectedRow> Mutations<SymbolType>::collectSelected(const Database& database, PendingScans& scans) const {
   std::vector<SelectedRow> selected;
   if (!scans.fetch && !scans.row_slot) {
      return selected;
   }
   const MutationTableLayout& layout = database.getMutationTableLayout<SymbolType>();
   constexpr uint32_t n_symbols = SymbolType::VALID_MUTATION_SYMBOLS.size();
   const uint32_t* table = nullptr;
   const silo_gpu_mutation_row* rows = nullptr;
   uint32_t n_selected = 0;
   if (scans.row_slot) {
      std::tie(rows, n_selected) = scans.row_slot.wait();
   } else {
      table = static_cast<const uint32_t*>(scans.fetch.wait());
   }
   Trace::mark("counts_on_host");
   if (scans.check_fetch) {
      checkSameQuery(database, static_cast<const uint32_t*>(scans.check_fetch.wait()), scans.fingerprint);
   }
   HostFetch whole_table;
   if (table == nullptr && n_selected > scans.row_capacity) {  // more rows than the list holds: take the whole table after all
      whole_table = HostFetch(scans.device_table.get(), scans.table_bytes, queryStream());
      table = static_cast<const uint32_t*>(whole_table.wait());
   }
   if (table != nullptr) {
      // the row selection of mutations.cpp:184-232 on the host, over the count table the device filled: counts[position][valid symbol]
      for (const auto& sequence_name : scans.sequence_names) {
         const SequenceStore<SymbolType>& sequence_store = database.getSequenceStores<SymbolType>().at(sequence_name);
         const uint32_t* counts = table + static_cast<size_t>(layout.position_offset.at(sequence_name)) * n_symbols;
         const auto& reference = sequence_store.reference_sequence;
         for (size_t position = 0; position < reference.size(); ++position) {
            const uint32_t* cell = counts + position * n_symbols;
            uint32_t covered = 0;  // rows of the filter with a valid symbol here; uint32 wrap-around like the reference's sum
            for (size_t k = 0; k < n_symbols; ++k) {
               covered += cell[k];
            }
            if (covered == 0) {
               continue;
            }
            // a symbol is reported when count >= ceil(covered * minProportion); written as the reference writes it (IEEE double,
            // then "count > that - 1" in uint32), because the rounding decides rows at the boundary
            const uint32_t must_exceed = min_proportion == 0 ? 0 : static_cast<uint32_t>(std::ceil(static_cast<double>(covered) * min_proportion) - 1);
            for (size_t k = 0; k < n_symbols; ++k) {
               if (SymbolType::VALID_MUTATION_SYMBOLS[k] != reference[position] && cell[k] > must_exceed) {
                  selected.push_back({&sequence_name, &sequence_store, static_cast<uint32_t>(position), static_cast<uint32_t>(k), cell[k], covered});
               }
            }
         }
      }
      return selected;
   }
   // the device appends in no particular order; the reference emits stores in request order, positions
   // ascending, symbols in VALID_MUTATION_SYMBOLS order (mutations.cpp:190-229, 259-268)
   std::vector<silo_gpu_mutation_row> sorted(rows, rows + n_selected);
   const auto before = [](const silo_gpu_mutation_row& a, const silo_gpu_mutation_row& b) {
      return a.position != b.position ? a.position < b.position : a.symbol_index < b.symbol_index;
   };
   std::sort(sorted.begin(), sorted.end(), before);
   selected.reserve(sorted.size());
   for (const auto& sequence_name : scans.sequence_names) {
      const SequenceStore<SymbolType>& sequence_store = database.getSequenceStores<SymbolType>().at(sequence_name);
      const uint32_t offset = layout.position_offset.at(sequence_name);
      const auto length = static_cast<uint32_t>(sequence_store.reference_sequence.size());
      auto row = std::lower_bound(sorted.begin(), sorted.end(), silo_gpu_mutation_row{offset, 0, 0, 0}, before);
      for (; row != sorted.end() && row->position < offset + length; ++row) {
         selected.push_back({&sequence_name, &sequence_store, row->position - offset, row->symbol_index, row->count, row->total});
      }
   }
   return selected;
}

template <typename SymbolType>
std::string Mutations<SymbolType>::mutationName(const SelectedRow& row) const {  // "<reference symbol><1-based position><symbol>", mutations.cpp:213-216
   const char from = SymbolType::symbolToChar(row.sequence_store->reference_sequence.at(row.position));
   const char to = SymbolType::symbolToChar(SymbolType::VALID_MUTATION_SYMBOLS.at(row.symbol_index));
   return from + std::to_string(row.position + 1) + to;
}

template <typename SymbolType>
QueryResult Mutations<SymbolType>::collect(const Database& database, PendingScans& scans) const {
   std::vector<QueryResultEntry> result_rows;
   const std::vector<SelectedRow> selected = collectSelected(database, scans);
   result_rows.reserve(selected.size());
   for (const SelectedRow& row : selected) {
      addSelectedRowToOutput(
         *row.sequence_name, *row.sequence_store, row.position, silo_gpu_mutation_row{row.position, row.symbol_index, row.count, row.total}, result_rows
      );
   }
   Trace::mark("rows_built");
   return QueryResult{std::move(result_rows)};
}

template <typename SymbolType>
QueryResult Mutations<SymbolType>::finish(const Database& database, Action::Pending& pending) const {
   return orderAndLimit(collect(database, dynamic_cast<PendingScans&>(pending)));
}

template <typename SymbolType>
std::string Mutations<SymbolType>::finishJson(const Database& database, Action::Pending& pending) const {
   // The bytes toJsonText(finish(...)) would give, written from the selected rows themselves.  Ordering: the comparator, the
   // index array and the std:: algorithms of Action::applySort over the same values (a field of a row is a string, an
   // int32 or a double there too), so ties fall the same way.
   const std::vector<SelectedRow> selected = collectSelected(database, dynamic_cast<PendingScans&>(pending));
   const size_t n_rows = selected.size();
   std::vector<uint32_t> order(n_rows);
   for (size_t row = 0; row < n_rows; ++row) {
      order[row] = static_cast<uint32_t>(row);
   }
   std::vector<std::string> names;  // only where a row's name is needed before the rows are written: ordered by "mutation"
   const size_t first = std::min<size_t>(offset.value_or(0), n_rows);
   const size_t count = limit.has_value() ? std::min<size_t>(*limit, n_rows - first) : n_rows - first;
   if (!order_by_fields.empty() && offset.value_or(0) < n_rows) {
      enum class Key { MUTATION, PROPORTION, COUNT };
      std::vector<std::pair<Key, bool>> keys;
      for (const OrderByField& field : order_by_fields) {
         keys.emplace_back(field.name == MUTATION_FIELD_NAME ? Key::MUTATION : (field.name == PROPORTION_FIELD_NAME ? Key::PROPORTION : Key::COUNT), field.ascending);
         if (keys.back().first == Key::MUTATION && names.empty()) {
            names.reserve(n_rows);
            for (const SelectedRow& row : selected) {
               names.push_back(mutationName(row));
            }
         }
      }
      const auto proportion = [&](uint32_t row) { return static_cast<double>(selected[row].count) / static_cast<double>(selected[row].total); };
      const auto before = [&](uint32_t left, uint32_t right) {
         for (const auto& [key, ascending] : keys) {
            if (key == Key::MUTATION) {
               if (!(names[left] == names[right])) {
                  return (names[left] < names[right]) == ascending;
               }
            } else if (key == Key::PROPORTION) {
               const double a = proportion(left), b = proportion(right);
               if (!(a == b)) {
                  return (a < b) == ascending;
               }
            } else {
               const auto a = static_cast<int32_t>(selected[left].count), b = static_cast<int32_t>(selected[right].count);
               if (!(a == b)) {
                  return (a < b) == ascending;
               }
            }
         }
         return false;
      };
      const size_t window_end = limit.has_value() ? std::min<size_t>(n_rows, static_cast<size_t>(*limit) + offset.value_or(0)) : n_rows;
      if (window_end < n_rows) {
         std::partial_sort(order.begin(), order.begin() + static_cast<std::ptrdiff_t>(window_end), order.end(), before);
      } else {
         std::sort(order.begin(), order.end(), before);
      }
   }
   std::string out;
   out.reserve(32 + count * 112);
   out += "{\"queryResult\":[";
   char digits[16];
   for (size_t k = 0; k < count; ++k) {
      const uint32_t index = order[first + k];
      const SelectedRow& row = selected[index];
      // keys in the order of the reference's std::map: count < mutation < proportion < sequenceName
      out += k == 0 ? "{\"count\":" : ",{\"count\":";
      out.append(digits, std::to_chars(digits, digits + sizeof(digits), static_cast<int32_t>(row.count)).ptr);
      out += ",\"mutation\":";
      json::Value::appendString(out, names.empty() ? mutationName(row) : names[index]);
      out += ",\"proportion\":";
      json::Value::appendDouble(out, static_cast<double>(row.count) / static_cast<double>(row.total));
      out += ",\"sequenceName\":";
      json::Value::appendString(out, *row.sequence_name);
      out.push_back('}');
   }
   out += "]}";
   Trace::mark("rows_built");
   return out;
}

template <typename SymbolType>
QueryResult Mutations<SymbolType>::execute(const Database& database, std::vector<OperatorResult> bitmap_filter) const {  // mutations.cpp:234-272
   // unordered result of the two phases run back to back (executeAndOrder applies the ordering)
   auto pending = begin(database, std::move(bitmap_filter));
   if (ScanBatcher* batcher = ScanBatcher::active(); batcher != nullptr) {
      batcher->flush();
   }
   return collect(database, dynamic_cast<PendingScans&>(*pending));
}

template class Mutations<Nucleotide>;
template class Mutations<AminoAcid>;

// ---- JSON -> Action -----------------------------------------------------------------------------------
namespace {

OrderByField parseOrderByField(const json::Value& json) {  // action.cpp:119-142
   if (json.is_string()) {
      return {json.as_string(), true};
   }
   const std::string message = "The orderByField '" + json.dump() +
                               "' must be either a string or an object containing the fields 'field':string and "
                               "'order':string, where the value of order is 'ascending' or 'descending'";
   CHECK_SILO_QUERY(
      json.is_object() && json.contains("field") && json.contains("order") && json["field"].is_string() && json["order"].is_string(), message
   )
   const std::string field_name = json["field"].as_string();
   const std::string order_string = json["order"].as_string();
   CHECK_SILO_QUERY(order_string == "ascending" || order_string == "descending", message)
   return {field_name, order_string == "ascending"};
}

/// A field that holds one name or an array of names (the sequenceName / column fields of Mutations, Insertions, Fasta
/// and FastaAligned): the names in order, none when the field is absent.  The wording of the two complaints belongs to the
/// action (the e2e fixtures assert it), so the caller supplies it.
std::vector<std::string> parseNames(
   const json::Value& json, std::string_view field, bool required, const std::string& wrong_type_message,
   const std::function<std::string(const json::Value&)>& wrong_element_message
) {
   const bool present = json.contains(field);
   CHECK_SILO_QUERY((present || !required) && (!present || json[field].is_string() || json[field].is_array()), wrong_type_message)
   std::vector<std::string> names;
   if (!present) {
      return names;
   }
   if (json[field].is_string()) {
      names.push_back(json[field].as_string());
      return names;
   }
   for (const auto& element : json[field].items()) {
      CHECK_SILO_QUERY(element.is_string(), wrong_element_message(element))
      names.push_back(element.as_string());
   }
   return names;
}

template <typename SymbolType>
std::unique_ptr<Action> parseMutations(const json::Value& json) {  // mutations.cpp:274-316
   std::vector<std::string> sequence_names = parseNames(
      json, "sequenceName", false, "Mutations action can have the field sequenceName of type string or an array of strings, but no other type",
      [](const json::Value& element) {
         return "The field sequenceName of Mutations action must have type string or an array, if present. Found:" + element.dump();
      }
   );
   CHECK_SILO_QUERY(
      json.contains("minProportion") && json["minProportion"].is_number(),
      "Mutations action must contain the field minProportion of type number with limits [0.0, 1.0]. Only mutations are returned if the "
      "proportion of sequences having this mutation, is at least minProportion"
   )
   const double min_proportion = json["minProportion"].as_double();
   CHECK_SILO_QUERY(min_proportion >= 0 && min_proportion <= 1, "Invalid proportion: minProportion must be in interval [0.0, 1.0]")
   return std::make_unique<Mutations<SymbolType>>(std::move(sequence_names), min_proportion);
}

template <typename SymbolType>
std::unique_ptr<Action> parseInsertions(const json::Value& json) {  // insertions.cpp:260-302
   std::vector<std::string> sequence_names = parseNames(
      json, "sequenceName", false, "Insertions action can have the field sequenceName of type string or an array of strings, but no other type",
      [](const json::Value& element) {
         return "The field sequenceName of the Insertions action must have type string or an array, if present. Found:" + element.dump();
      }
   );
   std::vector<std::string> column_names = parseNames(
      json, "column", false, "Insertions action can have the field column of type string or an array of strings, but no other type",
      [](const json::Value& element) {
         return "The field column of the Insertions action must have type string or an array, if present. Found:" + element.dump();
      }
   );
   return std::make_unique<InsertionAggregation<SymbolType>>(std::move(column_names), std::move(sequence_names));
}

/// The sequenceName field of Fasta / FastaAligned (fasta.cpp:247-270, fasta_aligned.cpp:138-161): required.
std::vector<std::string> parseRequiredSequenceNames(const json::Value& json, const std::string& action_name) {
   const std::string wrong_type = action_name + " action must have the field sequenceName of type string or an array of strings";
   return parseNames(json, "sequenceName", true, wrong_type, [&](const json::Value& element) {
      return wrong_type + "; while parsing array encountered the element " + element.dump() + " which is not of type string";
   });
}

}  // namespace

namespace {

std::vector<std::string> parseFieldList(const json::Value& json, std::string_view field) {
   std::vector<std::string> names;
   if (json.contains(field)) {
      for (const auto& name : json[field].items()) {
         names.push_back(name.as_string());
      }
   }
   return names;
}

using ActionParser = std::unique_ptr<Action> (*)(const json::Value&);
/// The "type" values of an action (action.cpp:144-187) and what builds each.
constexpr std::pair<std::string_view, ActionParser> ACTION_TYPES[] = {
   {"Aggregated", [](const json::Value& json) -> std::unique_ptr<Action> { return std::make_unique<Aggregated>(parseFieldList(json, "groupByFields")); }},
   {"Mutations", parseMutations<Nucleotide>},
   {"AminoAcidMutations", parseMutations<AminoAcid>},
   {"Details", [](const json::Value& json) -> std::unique_ptr<Action> { return std::make_unique<Details>(parseFieldList(json, "fields")); }},
   {"FastaAligned", [](const json::Value& json) -> std::unique_ptr<Action> { return std::make_unique<FastaAligned>(parseRequiredSequenceNames(json, "FastaAligned")); }},
   {"Fasta", [](const json::Value& json) -> std::unique_ptr<Action> { return std::make_unique<Fasta>(parseRequiredSequenceNames(json, "Fasta")); }},
   {"Insertions", parseInsertions<Nucleotide>},
   {"AminoAcidInsertions", parseInsertions<AminoAcid>},
};

}  // namespace

std::unique_ptr<Action> parseAction(const json::Value& json) {  // action.cpp:144-187
   CHECK_SILO_QUERY(json.contains("type"), "The field 'type' is required in any action")
   CHECK_SILO_QUERY(json["type"].is_string(), "The field 'type' in all actions needs to be a string, but is: " + json["type"].dump())
   const std::string& type = json["type"].as_string();
   std::unique_ptr<Action> action;
   for (const auto& [name, parser] : ACTION_TYPES) {
      if (type == name) {
         action = parser(json);
         break;
      }
   }
   if (action == nullptr) {
      throw QueryParseException(type + " is not a valid action");
   }
   std::vector<OrderByField> ordering;
   if (json.contains("orderByFields")) {
      for (const auto& field : json["orderByFields"].items()) {
         ordering.push_back(parseOrderByField(field));
      }
   }
   const auto optionalCount = [&](std::string_view field, const char* complaint) -> std::optional<uint32_t> {
      if (!json.contains(field)) {
         return std::nullopt;
      }
      CHECK_SILO_QUERY(json[field].is_number_unsigned(), complaint)
      return json[field].as_uint32();
   };
   const auto limit = optionalCount("limit", "If the action contains a limit, it must be a non-negative number");
   const auto offset = optionalCount("offset", "If the action contains an offset, it must be a non-negative number");
   action->setOrdering(ordering, limit, offset);
   return action;
}

}  // namespace silo::query_engine::actions
