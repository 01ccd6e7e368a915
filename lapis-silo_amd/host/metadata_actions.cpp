// metadata_actions.cpp — the actions that read metadata columns:
//   Aggregated with groupByFields   src/silo/query_engine/actions/aggregated.cpp:100-149
//   Details                         src/silo/query_engine/actions/details.cpp
//   FastaAligned                    src/silo/query_engine/actions/fasta_aligned.cpp
#include <algorithm>
#include <type_traits>
#include <unordered_map>

#include "query_engine.h"

namespace silo::query_engine::actions {

namespace {

using storage::column::MetadataColumnPartition;

const MetadataColumnPartition& columnOf(const DatabasePartition& partition, const std::string& name) {
   const auto found = partition.columns.metadata_columns.find(name);
   if (found == partition.columns.metadata_columns.end()) {
      throw std::runtime_error("the metadata column '" + name + "' was not loaded into this database");
   }
   return found->second;
}

/// Under position-range sharding every rank holds every row (and every metadata column), so row-wise actions simply
/// run on each rank.  A sequence-id shard is one DatabasePartition of the reference (query_engine.cpp:40-49): an action
/// whose rows are rows of the database (Details, Fasta, FastaAligned) answers for ITS rows and the front end concatenates
/// the shards' responses — no collective; ordering, offset and limit then hold per shard.  An action
/// that merges rows across partitions (Aggregated with groupByFields, Insertions) would need a gather that is not built.
void requireRowsMergedLocally(const Database& database, const char* what) {
   if (database.shard_world > 1 && !database.shard_by_position) {
      throw std::runtime_error(std::string(what) + " is not supported on a database sharded by sequence id");
   }
}
/// Needs every position of a sequence on this device: not on a position-range shard.
void requireAllPositionsLocal(const Database& database, const char* what) {
   if (database.shard_world > 1 && database.shard_by_position) {
      throw std::runtime_error(std::string(what) + " is not supported on a database sharded by position range");
   }
}

/// The set bits of a filter, fetched from the device.
std::vector<uint32_t> selectedRows(const DatabasePartition& partition, const OperatorResult& filter) {
   std::vector<uint32_t> rows;
   const uint32_t cardinality = filter.cardinality();
   if (cardinality == 0) {
      return rows;
   }
   rows.reserve(cardinality);
   std::vector<uint64_t> words(partition.rowWords());
   checkGpu(silo_gpu_memcpy_d2h(words.data(), filter.bitset(), words.size() * sizeof(uint64_t), queryStream()), "silo_gpu_memcpy_d2h");
   for (size_t word = 0; word < words.size(); ++word) {
      uint64_t bits = words[word];
      while (bits != 0) {
         rows.push_back(static_cast<uint32_t>(word * 64 + static_cast<uint32_t>(__builtin_ctzll(bits))));
         bits &= bits - 1;
      }
   }
   return rows;
}

}  // namespace

// ---- Aggregated with groupByFields ---------------------------------------------------------------------
QueryResult Aggregated::aggregateWithGrouping(const Database& database, std::vector<OperatorResult>& bitmap_filter) const {
   requireRowsMergedLocally(database, "Aggregated with groupByFields");
   struct Group {
      uint32_t count = 0;
      std::vector<JsonValue> values;
   };
   // Tuples are keyed by their raw bytes (tuple.cpp:389-391); std::map instead of the reference's unordered_map
   // makes the (unspecified) order of the result rows deterministic.
   std::map<std::string, Group> groups;
   const size_t n_fields = group_by_fields.size();

   struct InFlight {
      const DatabasePartition* partition;
      std::vector<const MetadataColumnPartition*> columns;
      std::vector<uint32_t> cardinalities;
      DeviceBuffer device_counts;
      size_t n_bins = 0;
      HostFetch fetch;
   };
   std::vector<InFlight> in_flight;
   for (size_t partition_id = 0; partition_id < database.partitions.size(); ++partition_id) {
      const DatabasePartition& partition = database.partitions[partition_id];
      OperatorResult& filter = bitmap_filter[partition_id];
      std::vector<const MetadataColumnPartition*> columns;
      for (const std::string& field : group_by_fields) {
         columns.push_back(&columnOf(partition, field));
      }
      if (partition.sequence_count == 0) {
         continue;
      }
      // dictionary ids per field: a tuple is a mixed-radix number, the histogram of those numbers is the group-by
      std::vector<const uint32_t*> ids;
      std::vector<uint32_t> cardinalities;
      uint64_t n_bins = 1;
      for (const MetadataColumnPartition* column : columns) {
         const MetadataColumnPartition::Groups column_groups = column->groups();
         ids.push_back(column_groups.device_ids);
         cardinalities.push_back(column_groups.cardinality);
         n_bins = std::min<uint64_t>(n_bins * std::max<uint32_t>(column_groups.cardinality, 1), uint64_t{1} << 40);
      }
      // the dense histogram comes back whole (4 bytes per potential tuple): beyond a million potential tuples the hash
      // table path, which returns only the tuples that occur, moves less data
      constexpr uint64_t DENSE_HISTOGRAM_LIMIT = uint64_t{1} << 20;
      static_assert(DENSE_HISTOGRAM_LIMIT <= SILO_GPU_MAX_GROUP_BINS);
      if (n_fields <= SILO_GPU_MAX_GROUP_COLUMNS && n_bins <= DENSE_HISTOGRAM_LIMIT) {
         InFlight& launch = in_flight.emplace_back();
         launch.partition = &partition;
         launch.columns = columns;
         launch.cardinalities = cardinalities;
         launch.n_bins = static_cast<size_t>(n_bins);
         launch.device_counts = partition.pool.acquire(launch.n_bins * sizeof(uint32_t));
         checkGpu(silo_gpu_memset_async(launch.device_counts.get(), 0, launch.n_bins * sizeof(uint32_t), queryStream()), "silo_gpu_memset_async");
         checkGpu(
            silo_gpu_group_count(
               partition.store, filter.bitset(), ids.data(), cardinalities.data(), static_cast<uint32_t>(n_fields),
               static_cast<uint32_t*>(launch.device_counts.get()), queryStream()
            ),
            "silo_gpu_group_count"
         );
         launch.fetch = HostFetch(launch.device_counts.get(), launch.n_bins * sizeof(uint32_t), queryStream());
         continue;
      }
      // a tuple space beyond the dense histogram: hash table in HBM keyed by the 64-bit tuple id (K6b)
      if (n_fields > SILO_GPU_MAX_GROUP_COLUMNS) {
         throw QueryCompilationException("Compilation Error: more than " + std::to_string(SILO_GPU_MAX_GROUP_COLUMNS) + " groupByFields");
      }
      const uint32_t selected = filter.cardinality();
      if (selected == 0) {
         continue;
      }
      uint64_t* device_keys = nullptr;
      uint32_t* device_group_counts = nullptr;
      uint32_t n_groups = 0;
      checkGpu(
         silo_gpu_group_count_hashed(
            partition.store, filter.bitset(), ids.data(), cardinalities.data(), static_cast<uint32_t>(n_fields), selected, &device_keys,
            &device_group_counts, &n_groups, queryStream()
         ),
         "silo_gpu_group_count_hashed"
      );
      std::vector<uint64_t> tuple_ids(n_groups);
      std::vector<uint32_t> tuple_counts(n_groups);
      int status = 0;
      if (n_groups != 0) {
         status = silo_gpu_memcpy_d2h(tuple_ids.data(), device_keys, n_groups * sizeof(uint64_t), queryStream());
         if (status == 0) {
            status = silo_gpu_memcpy_d2h(tuple_counts.data(), device_group_counts, n_groups * sizeof(uint32_t), queryStream());
         }
      }
      silo_gpu_free(device_keys);
      silo_gpu_free(device_group_counts);
      checkGpu(status, "silo_gpu_memcpy_d2h");
      std::vector<uint32_t> digits(n_fields);
      std::string key;
      for (uint32_t group = 0; group < n_groups; ++group) {
         uint64_t rest = tuple_ids[group];  // first field most significant
         for (size_t field = n_fields; field-- > 0;) {
            digits[field] = static_cast<uint32_t>(rest % cardinalities[field]);
            rest /= cardinalities[field];
         }
         key.clear();
         for (size_t field = 0; field < n_fields; ++field) {
            columns[field]->appendKeyOfGroup(digits[field], key);
         }
         Group& entry = groups[key];
         if (entry.count == 0) {
            for (size_t field = 0; field < n_fields; ++field) {
               entry.values.push_back(columns[field]->jsonOfGroup(digits[field]));
            }
         }
         entry.count += tuple_counts[group];
      }
   }
   Trace::mark("groups_launched");
   for (InFlight& launch : in_flight) {
      const auto* counts = static_cast<const uint32_t*>(launch.fetch.wait());
      std::vector<uint32_t> digits(n_fields);
      std::string key;
      for (size_t bin = 0; bin < launch.n_bins; ++bin) {
         if (counts[bin] == 0) {
            continue;
         }
         size_t rest = bin;  // first field most significant
         for (size_t field = n_fields; field-- > 0;) {
            digits[field] = static_cast<uint32_t>(rest % launch.cardinalities[field]);
            rest /= launch.cardinalities[field];
         }
         key.clear();
         for (size_t field = 0; field < n_fields; ++field) {
            launch.columns[field]->appendKeyOfGroup(digits[field], key);
         }
         Group& group = groups[key];
         if (group.count == 0) {
            for (size_t field = 0; field < n_fields; ++field) {
               group.values.push_back(launch.columns[field]->jsonOfGroup(digits[field]));
            }
         }
         group.count += counts[bin];
      }
   }
   Trace::mark("groups_on_host");
   QueryResult result;  // generateResult, aggregated.cpp:44-56
   result.query_result.reserve(groups.size());
   for (auto& [key, group] : groups) {
      QueryResultEntry& entry = result.query_result.emplace_back();
      for (size_t field = 0; field < n_fields; ++field) {
         entry.fields[group_by_fields[field]] = std::move(group.values[field]);
      }
      entry.fields["count"] = static_cast<int32_t>(group.count);
   }
   return result;
}

// ---- Details ---------------------------------------------------------------------------------------------
namespace {

std::vector<storage::ColumnMetadata> parseFields(const Database& database, const std::vector<std::string>& fields) {  // details.cpp:22-35
   if (fields.empty()) {
      return database.database_config.metadata;
   }
   std::vector<storage::ColumnMetadata> field_metadata;
   for (const std::string& field : fields) {
      const auto metadata = database.database_config.getMetadata(field);
      CHECK_SILO_QUERY(metadata.has_value(), "Metadata field " + field + " not found.")
      field_metadata.push_back(*metadata);
   }
   return field_metadata;
}

}  // namespace

void Details::validateOrderByFields(const Database& database) const {  // details.cpp:43-59
   const std::vector<storage::ColumnMetadata> field_metadata = parseFields(database, fields);
   for (const OrderByField& field : order_by_fields) {
      CHECK_SILO_QUERY(
         std::any_of(field_metadata.begin(), field_metadata.end(), [&](const storage::ColumnMetadata& metadata) { return metadata.name == field.name; }),
         "OrderByField " + field.name + " is not contained in the result of this operation."
      )
   }
}

QueryResult Details::execute(const Database& /*database*/, std::vector<OperatorResult> /*bitmap_filter*/) const {
   return QueryResult{};  // details.cpp:61-66: everything happens in executeAndOrder
}

QueryResult Details::finish(const Database& database, Pending& pending) const {
   return executeAndOrder(database, std::move(pending.bitmap_filter));
}

QueryResult Details::executeAndOrder(const Database& database, std::vector<OperatorResult> bitmap_filter) const {  // details.cpp:186-219
   validateOrderByFields(database);
   const std::vector<storage::ColumnMetadata> field_metadata = parseFields(database, fields);

   struct Row {
      uint32_t partition;
      uint32_t row;
   };
   std::vector<Row> tuples;
   // columns of every partition, in field order (the TupleFactory of a partition)
   std::vector<std::vector<const storage::column::MetadataColumnPartition*>> columns(database.partitions.size());
   for (size_t partition_id = 0; partition_id < database.partitions.size(); ++partition_id) {
      const DatabasePartition& partition = database.partitions[partition_id];
      for (const auto& metadata : field_metadata) {
         columns[partition_id].push_back(&columnOf(partition, metadata.name));
      }
      for (const uint32_t row : selectedRows(partition, bitmap_filter[partition_id])) {
         tuples.push_back({static_cast<uint32_t>(partition_id), row});
      }
   }
   Trace::mark("rows_selected");

   // Tuple::compareLess (tuple.cpp:372-387) over the orderByFields, on the raw column values
   struct CompareField {
      size_t index;
      bool ascending;
   };
   std::vector<CompareField> compare_fields;
   for (const OrderByField& order_by : order_by_fields) {
      for (size_t index = 0; index < field_metadata.size(); ++index) {
         if (field_metadata[index].name == order_by.name) {
            compare_fields.push_back({index, order_by.ascending});
            break;
         }
      }
   }
   const auto less = [&](const Row& a, const Row& b) {
      for (const CompareField& field : compare_fields) {
         const int compared = columns[a.partition][field.index]->compareRows(a.row, *columns[b.partition][field.index], b.row);
         if (compared < 0) {
            return field.ascending;
         }
         if (compared > 0) {
            return !field.ascending;
         }
      }
      return false;
   };
   // With a limit the reference keeps the `limit + offset` smallest tuples per partition in a heap and merges
   // them (:88-147).  Its heap is offered the first row past the prefix twice (:118-134), which can duplicate a
   // row in the result; that defect is not reproduced: the smallest `limit + offset` tuples, each once.
   size_t to_produce = tuples.size();
   if (limit.has_value()) {
      to_produce = std::min<size_t>(tuples.size(), static_cast<size_t>(limit.value()) + offset.value_or(0));
   }
   if (!compare_fields.empty()) {
      if (to_produce < tuples.size()) {
         std::partial_sort(tuples.begin(), tuples.begin() + static_cast<int64_t>(to_produce), tuples.end(), less);
      } else {
         std::sort(tuples.begin(), tuples.end(), less);
      }
   }
   tuples.resize(to_produce);

   // only the rows that survive offset / limit are rendered (applyOffsetAndLimit, action.cpp:68-91)
   const size_t begin = std::min<size_t>(offset.value_or(0), tuples.size());
   QueryResult results_in_format;
   results_in_format.query_result.reserve(tuples.size() - begin);
   for (size_t index = begin; index < tuples.size(); ++index) {
      const Row& tuple = tuples[index];
      QueryResultEntry& entry = results_in_format.query_result.emplace_back();
      for (size_t field = 0; field < field_metadata.size(); ++field) {
         entry.fields[field_metadata[field].name] = columns[tuple.partition][field]->jsonOfRow(tuple.row);
      }
   }
   Trace::mark("rows_built");
   return results_in_format;
}

// ---- Fasta (fasta.cpp) ---------------------------------------------------------------------------------------
void Fasta::validateOrderByFields(const Database& database) const {  // :42-57
   const std::string& primary_key_field = database.database_config.primary_key;
   for (const OrderByField& field : order_by_fields) {
      std::string joined;
      for (size_t i = 0; i < sequence_names.size(); ++i) {
         joined += (i == 0 ? "" : ",") + sequence_names[i];
      }
      CHECK_SILO_QUERY(
         field.name == primary_key_field || std::find(sequence_names.begin(), sequence_names.end(), field.name) != sequence_names.end(),
         "The only fields returned by the Fasta action are " + joined + " and " + primary_key_field
      )
   }
}

QueryResult Fasta::execute(const Database& database, std::vector<OperatorResult> bitmap_filter) const {  // :214-245
   for (const std::string& sequence_name : sequence_names) {
      // every nucleotide sequence has an unaligned store (database.cpp:664-673)
      CHECK_SILO_QUERY(
         database.nuc_sequences.count(sequence_name) != 0, "Database does not contain an unaligned sequence with name: '" + sequence_name + "'"
      )
   }
   const std::string& primary_key_column = database.database_config.primary_key;
   size_t total_count = 0;
   for (const auto& filter : bitmap_filter) {
      total_count += filter.cardinality();
   }
   CHECK_SILO_QUERY(total_count <= SEQUENCE_LIMIT, "Fasta action currently limited to " + std::to_string(SEQUENCE_LIMIT) + " sequences")
   QueryResult results;
   results.query_result.reserve(total_count);
   for (size_t partition_id = 0; partition_id < database.partitions.size(); ++partition_id) {
      const DatabasePartition& partition = database.partitions[partition_id];
      const std::vector<uint32_t> rows = selectedRows(partition, bitmap_filter[partition_id]);
      if (rows.empty()) {
         continue;
      }
      const MetadataColumnPartition& primary_key = columnOf(partition, primary_key_column);
      for (const uint32_t row : rows) {
         QueryResultEntry& entry = results.query_result.emplace_back();
         JsonValue key = primary_key.jsonOfRow(row);
         if (!key.has_value()) {
            throw std::runtime_error("Detected primary_key in column '" + primary_key_column + "' that is null.");
         }
         entry.fields.emplace(primary_key_column, std::move(key));
         for (const std::string& sequence_name : sequence_names) {
            const auto found = partition.unaligned_nuc_sequences.find(sequence_name);
            if (found != partition.unaligned_nuc_sequences.end() && row < found->second.size() && found->second[row].has_value()) {
               entry.fields.emplace(sequence_name, *found->second[row]);
            } else {
               entry.fields.emplace(sequence_name, std::nullopt);
            }
         }
      }
   }
   return results;
}

// ---- Insertions / AminoAcidInsertions (insertions.cpp) ---------------------------------------------------
template <typename SymbolType>
void InsertionAggregation<SymbolType>::validateOrderByFields(const Database& /*database*/) const {  // :41-59
   for (const OrderByField& field : order_by_fields) {
      CHECK_SILO_QUERY(
         field.name == "position" || field.name == "insertions" || field.name == "sequenceName" || field.name == "count",
         "OrderByField " + field.name + " is not contained in the result of this operation."
      )
   }
}

template <typename SymbolType>
QueryResult InsertionAggregation<SymbolType>::execute(const Database& database, std::vector<OperatorResult> bitmap_filter) const {  // :126-258
   const config::ColumnType wanted_type =
      std::is_same_v<SymbolType, Nucleotide> ? config::ColumnType::NUC_INSERTION : config::ColumnType::AA_INSERTION;
   for (const std::string& column_name : column_names) {  // validateDatabaseColumnNames
      const auto metadata = database.database_config.getMetadata(column_name);
      CHECK_SILO_QUERY(
         metadata.has_value() && metadata->type == wanted_type,
         "The database does not contain the " + std::string(SymbolType::SYMBOL_NAME) + " column '" + column_name + "'"
      )
   }
   for (const std::string& sequence_name : sequence_names) {  // validateSequenceNames
      CHECK_SILO_QUERY(
         database.getSequenceStores<SymbolType>().count(sequence_name) != 0,
         "The database does not contain the " + std::string(SymbolType::SYMBOL_NAME) + " sequence '" + sequence_name + "'"
      )
   }
   requireRowsMergedLocally(database, "Insertions");

   // One k_count_pairs launch per (partition, column, sequence): the and_cardinality of the filter with the rows of
   // every distinct insertion at once (:196-206).  Launch all, then fetch.
   struct InFlight {
      const std::string* sequence_name;
      const storage::column::InsertionColumnPartition::SequenceIndex* index;
      DeviceBuffer device_counts;
      HostFetch fetch;
   };
   std::vector<InFlight> in_flight;
   for (size_t partition_id = 0; partition_id < database.partitions.size(); ++partition_id) {
      const DatabasePartition& partition = database.partitions[partition_id];
      const auto& insertion_columns = partition.columns.getInsertionColumns<SymbolType>();
      for (const std::string& column_name : column_names) {  // validatePartitionColumnNames
         CHECK_SILO_QUERY(
            insertion_columns.count(column_name) != 0,
            "The database does not contain the " + std::string(SymbolType::SYMBOL_NAME) + " column '" + column_name + "'"
         )
      }
      OperatorResult& filter = bitmap_filter[partition_id];
      if (filter.cardinality() == 0) {
         continue;
      }
      for (const auto& [column_name, insertion_column] : insertion_columns) {
         if (!column_names.empty() && std::find(column_names.begin(), column_names.end(), column_name) == column_names.end()) {
            continue;
         }
         for (const auto& [sequence_name, index] : insertion_column.getInsertionIndexes()) {
            if (!sequence_names.empty() && std::find(sequence_names.begin(), sequence_names.end(), sequence_name) == sequence_names.end()) {
               continue;
            }
            if (index.insertions.empty()) {
               continue;
            }
            InFlight& launch = in_flight.emplace_back();
            launch.sequence_name = &sequence_name;
            launch.index = &index;
            const size_t bytes = index.insertions.size() * sizeof(uint32_t);
            launch.device_counts = partition.pool.acquire(bytes);
            checkGpu(silo_gpu_memset_async(launch.device_counts.get(), 0, bytes, queryStream()), "silo_gpu_memset_async");
            checkGpu(
               silo_gpu_count_pairs(
                  partition.store, filter.bitset(), index.device_rows, index.device_ids, static_cast<uint32_t>(index.pair_rows.size()),
                  static_cast<uint32_t*>(launch.device_counts.get()), queryStream()
               ),
               "silo_gpu_count_pairs"
            );
            launch.fetch = HostFetch(launch.device_counts.get(), bytes, queryStream());
         }
      }
   }
   // sequence name -> (position, insertion) -> count; the reference's unordered_maps leave the row order unspecified
   std::map<std::string, std::map<std::pair<uint32_t, std::string>, uint32_t>> all_insertions;
   for (const InFlight& launch : in_flight) {
      const auto* counts = static_cast<const uint32_t*>(launch.fetch.wait());
      auto& per_sequence = all_insertions[*launch.sequence_name];
      for (size_t id = 0; id < launch.index->insertions.size(); ++id) {
         if (counts[id] > 0) {
            per_sequence[{launch.index->positions[id], launch.index->insertions[id]}] += counts[id];
         }
      }
   }
   QueryResult result;
   for (const auto& [sequence_name, per_sequence] : all_insertions) {
      for (const auto& [position_and_insertion, count] : per_sequence) {
         QueryResultEntry& entry = result.query_result.emplace_back();
         entry.fields["position"] = static_cast<int32_t>(position_and_insertion.first);
         entry.fields["sequenceName"] = sequence_name;
         entry.fields["insertions"] = position_and_insertion.second;
         entry.fields["count"] = static_cast<int32_t>(count);
      }
   }
   return result;
}

template class InsertionAggregation<Nucleotide>;
template class InsertionAggregation<AminoAcid>;

// ---- FastaAligned ----------------------------------------------------------------------------------------
void FastaAligned::validateOrderByFields(const Database& database) const {  // fasta_aligned.cpp:28-42
   const std::string& primary_key_field = database.database_config.primary_key;
   for (const OrderByField& field : order_by_fields) {
      std::string joined;
      for (size_t i = 0; i < sequence_names.size(); ++i) {
         joined += (i == 0 ? "" : ",") + sequence_names[i];
      }
      CHECK_SILO_QUERY(
         field.name == primary_key_field || std::find(sequence_names.begin(), sequence_names.end(), field.name) != sequence_names.end(),
         "The only fields returned by the FastaAligned action are " + joined + " and " + primary_key_field
      )
   }
}

QueryResult FastaAligned::execute(const Database& database, std::vector<OperatorResult> bitmap_filter) const {  // fasta_aligned.cpp:85-136
   struct Requested {
      std::string name;
      bool is_amino_acid;
   };
   std::vector<Requested> requested;  // nucleotide sequences first, then genes (:89-103); the row is a map anyway
   for (const std::string& sequence_name : sequence_names) {
      CHECK_SILO_QUERY(
         database.nuc_sequences.count(sequence_name) != 0 || database.aa_sequences.count(sequence_name) != 0,
         "Database does not contain a sequence with name: '" + sequence_name + "'"
      )
      requested.push_back({sequence_name, database.nuc_sequences.count(sequence_name) == 0});
   }
   size_t total_count = 0;
   for (const auto& filter : bitmap_filter) {
      total_count += filter.cardinality();
   }
   CHECK_SILO_QUERY(total_count < 10001, "FastaAligned action currently limited to 10000 sequences")
   requireAllPositionsLocal(database, "FastaAligned");

   const std::string& primary_key_column = database.database_config.primary_key;
   QueryResult results;
   for (size_t partition_id = 0; partition_id < database.partitions.size(); ++partition_id) {
      const DatabasePartition& partition = database.partitions[partition_id];
      const std::vector<uint32_t> rows = selectedRows(partition, bitmap_filter[partition_id]);
      if (rows.empty()) {
         continue;
      }
      const size_t first_entry = results.query_result.size();
      const MetadataColumnPartition& primary_key = columnOf(partition, primary_key_column);
      for (const uint32_t row : rows) {
         results.query_result.emplace_back().fields.emplace(primary_key_column, primary_key.jsonOfRow(row));
      }
      // reconstructSequence (:44-83) for all selected rows of a store at once: one gather over the planes
      DeviceBuffer device_rows = partition.pool.acquire(rows.size() * sizeof(uint32_t));
      checkGpu(silo_gpu_memcpy_h2d(device_rows.get(), rows.data(), rows.size() * sizeof(uint32_t), queryStream()), "silo_gpu_memcpy_h2d");
      for (const Requested& sequence : requested) {
         uint32_t seqstore_id = 0;
         size_t length = 0;
         if (sequence.is_amino_acid) {
            const auto& store = partition.aa_sequences.at(sequence.name);
            seqstore_id = store.seqstore_id;
            length = store.reference_sequence.size();
         } else {
            const auto& store = partition.nuc_sequences.at(sequence.name);
            seqstore_id = store.seqstore_id;
            length = store.reference_sequence.size();
         }
         std::vector<char> chars(rows.size() * length);
         if (!chars.empty()) {
            DeviceBuffer device_chars = partition.pool.acquire(chars.size());
            checkGpu(
               silo_gpu_reconstruct_sequences(
                  partition.store, seqstore_id, static_cast<const uint32_t*>(device_rows.get()), static_cast<uint32_t>(rows.size()),
                  static_cast<char*>(device_chars.get()), queryStream()
               ),
               "silo_gpu_reconstruct_sequences"
            );
            checkGpu(silo_gpu_memcpy_d2h(chars.data(), device_chars.get(), chars.size(), queryStream()), "silo_gpu_memcpy_d2h");
         }
         for (size_t index = 0; index < rows.size(); ++index) {
            results.query_result[first_entry + index].fields.emplace(sequence.name, std::string(chars.data() + index * length, length));
         }
      }
   }
   return results;
}

}  // namespace silo::query_engine::actions
