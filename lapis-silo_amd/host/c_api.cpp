// c_api.cpp — extern "C" surface of the C++ host (include/silo_engine.h).
#include <cstdlib>
#include <cstring>
#include <atomic>
#include <chrono>
#include <new>
#include <string>
#include <thread>
#include <vector>

#include "dataset_loader.h"
#include "query_engine.h"
#include "silo_engine.h"

struct silo_engine {
   silo::Database database;
};

namespace {

thread_local std::string g_error;

int fail(int code, const std::string& message) {
   g_error = message;
   return code;
}

char* duplicate(const std::string& text) {
   char* out = static_cast<char*>(std::malloc(text.size() + 1));
   if (out != nullptr) {
      std::memcpy(out, text.c_str(), text.size() + 1);
   }
   return out;
}

std::string errorDocument(const char* error, const std::string& message) {  // src/silo_api/error_request_handler.h ErrorResponse
   silo::json::Value doc = silo::json::Value::object();
   doc.set("error", silo::json::Value(error));
   doc.set("message", silo::json::Value(message));
   return doc.dump();
}

silo::DatabasePartition* partitionOf(silo_engine* engine, int partition) {
   if (engine == nullptr || partition < 0 || static_cast<size_t>(partition) >= engine->database.partitions.size()) {
      return nullptr;
   }
   return &engine->database.partitions[static_cast<size_t>(partition)];
}

template <typename Function>
int guarded(Function&& function) {
   try {
      return function();
   } catch (const std::exception& ex) {
      return fail(SILO_GPU_ERR_INVALID_ARGUMENT, ex.what());
   } catch (...) {
      return fail(SILO_GPU_ERR_INVALID_ARGUMENT, "unknown exception");
   }
}

int seqstoreId(const silo::DatabasePartition& partition, const char* sequence_name, int is_amino_acid) {
   if (sequence_name == nullptr) {
      return -1;
   }
   if (is_amino_acid != 0) {
      const auto found = partition.aa_sequences.find(sequence_name);
      return found == partition.aa_sequences.end() ? -1 : static_cast<int>(found->second.seqstore_id);
   }
   const auto found = partition.nuc_sequences.find(sequence_name);
   return found == partition.nuc_sequences.end() ? -1 : static_cast<int>(found->second.seqstore_id);
}

}  // namespace

extern "C" {

const char* silo_engine_last_error(void) {
   return g_error.c_str();
}

int silo_engine_create(
   const char* reference_genomes_json, const char* alias_json, const char* default_nucleotide_sequence, int device, silo_engine** out
) {
   if (reference_genomes_json == nullptr || out == nullptr) {
      return fail(SILO_GPU_ERR_INVALID_ARGUMENT, "silo_engine_create: null argument");
   }
   *out = nullptr;
   return guarded([&] {
      auto engine = std::make_unique<silo_engine>();
      engine->database.device = device;
      silo::setEngineDevice(device);
      engine->database.setReferenceGenomes(silo::json::parse(reference_genomes_json));
      if (alias_json != nullptr) {
         engine->database.alias_key = silo::PangoLineageAliasLookup::fromJson(silo::json::parse(alias_json));
      }
      if (default_nucleotide_sequence != nullptr) {
         engine->database.database_config.default_nucleotide_sequence = default_nucleotide_sequence;
      }
      *out = engine.release();
      return 0;
   });
}

void silo_engine_destroy(silo_engine* engine) {
   delete engine;
}

int silo_engine_create_from_directory(const char* directory, int device, silo_engine** out, char** out_summary_json) {
   if (directory == nullptr || out == nullptr) {
      return fail(SILO_GPU_ERR_INVALID_ARGUMENT, "silo_engine_create_from_directory: null argument");
   }
   *out = nullptr;
   return guarded([&] {
      auto engine = std::make_unique<silo_engine>();
      engine->database.device = device;
      silo::setEngineDevice(device);
      const auto summary = silo::preprocessing::loadDataset(engine->database, directory);
      if (out_summary_json != nullptr) {
         silo::json::Value doc = silo::json::Value::object();
         doc.set("sequenceCount", silo::json::Value(static_cast<uint64_t>(summary.sequence_count)));
         doc.set("nucleotideStores", silo::json::Value(static_cast<uint64_t>(summary.nucleotide_stores)));
         doc.set("aminoAcidStores", silo::json::Value(static_cast<uint64_t>(summary.amino_acid_stores)));
         doc.set("lineageColumns", silo::json::Value(static_cast<uint64_t>(summary.lineage_columns)));
         doc.set("nullSequences", silo::json::Value(static_cast<uint64_t>(summary.null_sequences)));
         *out_summary_json = duplicate(doc.dump());
      }
      *out = engine.release();
      return 0;
   });
}

int silo_engine_add_partition(silo_engine* engine, uint32_t sequence_count) {
   if (engine == nullptr) {
      return fail(SILO_GPU_ERR_INVALID_ARGUMENT, "silo_engine_add_partition: null engine");
   }
   return guarded([&] {
      engine->database.addPartition(sequence_count);
      return static_cast<int>(engine->database.partitions.size() - 1);
   });
}

int silo_engine_append_sequences(
   silo_engine* engine, int partition, const char* sequence_name, int is_amino_acid, uint32_t first_sequence, uint32_t n_sequences,
   const char* chars, const uint8_t* is_null
) {
   silo::DatabasePartition* part = partitionOf(engine, partition);
   if (part == nullptr) {
      return fail(SILO_GPU_ERR_INVALID_ARGUMENT, "no such partition");
   }
   const int id = seqstoreId(*part, sequence_name, is_amino_acid);
   if (id < 0) {
      return fail(SILO_GPU_ERR_INVALID_ARGUMENT, "no such sequence store");
   }
   const int status = silo_gpu_store_append_sequences(part->store, static_cast<uint32_t>(id), first_sequence, n_sequences, chars, is_null);
   return status == 0 ? 0 : fail(status, silo_gpu_last_error());
}

int silo_engine_build_pass(silo_engine* engine, int partition, const char* sequence_name, int is_amino_acid, int pass) {
   silo::DatabasePartition* part = partitionOf(engine, partition);
   if (part == nullptr) {
      return fail(SILO_GPU_ERR_INVALID_ARGUMENT, "no such partition");
   }
   const int id = seqstoreId(*part, sequence_name, is_amino_acid);
   if (id < 0) {
      return fail(SILO_GPU_ERR_INVALID_ARGUMENT, "no such sequence store");
   }
   const int status = silo_gpu_store_build_pass(part->store, static_cast<uint32_t>(id), pass);
   return status == 0 ? 0 : fail(status, silo_gpu_last_error());
}

int silo_engine_generate_synthetic(silo_engine* engine, int partition, const char* sequence_name, int is_amino_acid, const silo_gpu_synth_desc* synth) {
   silo::DatabasePartition* part = partitionOf(engine, partition);
   if (part == nullptr) {
      return fail(SILO_GPU_ERR_INVALID_ARGUMENT, "no such partition");
   }
   const int id = seqstoreId(*part, sequence_name, is_amino_acid);
   if (id < 0) {
      return fail(SILO_GPU_ERR_INVALID_ARGUMENT, "no such sequence store");
   }
   int status = 0;
   // The one-pass build holds 3 / 5 identity planes per position, the plane of the missing symbol and then the finished store
   // (about 2 more rows per position with its keys): where that does not fit the device, the store is built in two passes.
   bool two_pass = engine->database.two_pass_build;
   uint64_t free_bytes = 0, total_bytes = 0;
   if (!two_pass && silo_gpu_store_memory_info(part->store, &free_bytes, &total_bytes) == 0) {
      const uint64_t row_bytes = static_cast<uint64_t>(part->rowWords()) * sizeof(uint64_t);
      uint64_t positions = 0;  // of this rank's slice of the store
      if (is_amino_acid != 0) {
         const auto found = part->aa_sequences.find(sequence_name);
         positions = found != part->aa_sequences.end() ? found->second.position_end - found->second.position_begin : 0;
      } else {
         const auto found = part->nuc_sequences.find(sequence_name);
         positions = found != part->nuc_sequences.end() ? found->second.position_end - found->second.position_begin : 0;
      }
      two_pass = (is_amino_acid != 0 ? 5u + 1u + 2u : 3u + 1u + 2u) * positions * row_bytes + (uint64_t{2} << 30) > free_bytes;
   }
   if (two_pass) {  // count, choose the layout of every position, then generate straight into it
      status = silo_gpu_store_build_pass(part->store, static_cast<uint32_t>(id), 1);
      status = status != 0 ? status : silo_gpu_store_generate_synthetic(part->store, static_cast<uint32_t>(id), synth);
      status = status != 0 ? status : silo_gpu_store_build_pass(part->store, static_cast<uint32_t>(id), 2);
   }
   status = status != 0 ? status : silo_gpu_store_generate_synthetic(part->store, static_cast<uint32_t>(id), synth);
   if (status == 0) {
      // the generator fills the whole sequence store in this one call: re-encode it now, so that the build-time planes of
      // the stores of a partition (112 GB for the nucleotide genome at 10 M sequences) are never resident together
      status = silo_gpu_store_finalize_seqstore(part->store, static_cast<uint32_t>(id));
   }
   return status == 0 ? 0 : fail(status, silo_gpu_last_error());
}

int silo_engine_set_lineage_column(silo_engine* engine, int partition, const char* column, const char* const* values, uint32_t n_rows) {
   silo::DatabasePartition* part = partitionOf(engine, partition);
   if (part == nullptr || column == nullptr || (values == nullptr && n_rows > 0)) {
      return fail(SILO_GPU_ERR_INVALID_ARGUMENT, "silo_engine_set_lineage_column: bad arguments");
   }
   if (n_rows != part->sequence_count) {
      return fail(SILO_GPU_ERR_INVALID_ARGUMENT, "lineage column must have one value per row");
   }
   return guarded([&] {
      part->columns.pango_lineage_columns.erase(column);
      auto& col = part->columns.pango_lineage_columns
                     .emplace(std::piecewise_construct, std::forward_as_tuple(column), std::forward_as_tuple(engine->database.alias_key, *part))
                     .first->second;
      for (uint32_t row = 0; row < n_rows; ++row) {
         if (values[row] == nullptr) {
            col.insertNull();
         } else {
            col.insert(values[row]);
         }
      }
      return 0;
   });
}

int silo_engine_set_lineage_column_ids(
   silo_engine* engine, int partition, const char* column, const char* const* dictionary, uint32_t n_dictionary, const uint32_t* value_ids,
   uint32_t n_rows
) {
   silo::DatabasePartition* part = partitionOf(engine, partition);
   if (part == nullptr || column == nullptr || dictionary == nullptr || value_ids == nullptr) {
      return fail(SILO_GPU_ERR_INVALID_ARGUMENT, "silo_engine_set_lineage_column_ids: bad arguments");
   }
   if (n_rows != part->sequence_count) {
      return fail(SILO_GPU_ERR_INVALID_ARGUMENT, "lineage column must have one value per row");
   }
   return guarded([&] {
      std::vector<std::string> names;
      names.reserve(n_dictionary);
      for (uint32_t i = 0; i < n_dictionary; ++i) {
         names.emplace_back(dictionary[i]);
      }
      part->columns.pango_lineage_columns.erase(column);
      auto& col = part->columns.pango_lineage_columns
                     .emplace(std::piecewise_construct, std::forward_as_tuple(column), std::forward_as_tuple(engine->database.alias_key, *part))
                     .first->second;
      col.setValues(std::move(names), value_ids, n_rows);
      return 0;
   });
}

int silo_engine_set_schema(silo_engine* engine, const char* primary_key, const char* date_to_sort_by) {
   if (engine == nullptr || primary_key == nullptr) {
      return fail(SILO_GPU_ERR_INVALID_ARGUMENT, "silo_engine_set_schema: null argument");
   }
   engine->database.database_config.primary_key = primary_key;
   if (date_to_sort_by != nullptr && date_to_sort_by[0] != '\0') {
      engine->database.database_config.date_to_sort_by = date_to_sort_by;
   } else {
      engine->database.database_config.date_to_sort_by.reset();
   }
   return 0;
}

int silo_engine_append_metadata(
   silo_engine* engine, int partition, const char* column, const char* column_type, const char* const* values, uint32_t n_values
) {
   silo::DatabasePartition* part = partitionOf(engine, partition);
   if (part == nullptr || column == nullptr || column_type == nullptr || (values == nullptr && n_values > 0)) {
      return fail(SILO_GPU_ERR_INVALID_ARGUMENT, "silo_engine_append_metadata: bad arguments");
   }
   std::optional<silo::config::ColumnType> type;
   if (std::strcmp(column_type, "indexed_string") == 0) {
      type = silo::config::ColumnType::INDEXED_STRING;
   } else {
      type = silo::config::columnTypeFromConfig(column_type, false);
   }
   if (!type.has_value()) {
      return fail(SILO_GPU_ERR_INVALID_ARGUMENT, std::string("silo_engine_append_metadata: unknown column type ") + column_type);
   }
   return guarded([&] {
      std::vector<std::string> texts;
      texts.reserve(n_values);
      for (uint32_t row = 0; row < n_values; ++row) {
         texts.emplace_back(values[row] == nullptr ? "" : values[row]);
      }
      engine->database.appendMetadata(*part, column, *type, texts);
      return 0;
   });
}

int silo_engine_append_unaligned_sequences(
   silo_engine* engine, int partition, const char* sequence_name, const char* const* sequences, uint32_t n_sequences
) {
   silo::DatabasePartition* part = partitionOf(engine, partition);
   if (part == nullptr || sequence_name == nullptr || (sequences == nullptr && n_sequences > 0)) {
      return fail(SILO_GPU_ERR_INVALID_ARGUMENT, "silo_engine_append_unaligned_sequences: bad arguments");
   }
   return guarded([&] {
      std::vector<std::optional<std::string>> values;
      values.reserve(n_sequences);
      for (uint32_t row = 0; row < n_sequences; ++row) {
         if (sequences[row] == nullptr) {
            values.emplace_back(std::nullopt);
         } else {
            values.emplace_back(std::string(sequences[row]));
         }
      }
      engine->database.appendUnalignedSequences(*part, sequence_name, std::move(values));
      return 0;
   });
}

int silo_engine_finalize(silo_engine* engine) {
   if (engine == nullptr) {
      return fail(SILO_GPU_ERR_INVALID_ARGUMENT, "silo_engine_finalize: null engine");
   }
   return guarded([&] {
      engine->database.finalize();
      return 0;
   });
}

int silo_engine_set_sharding(
   silo_engine* engine, uint32_t rank, uint32_t world, int shard_by_position, silo_engine_all_reduce_u32 all_reduce, void* context
) {
   if (engine == nullptr || world == 0 || rank >= world) {
      return fail(SILO_GPU_ERR_INVALID_ARGUMENT, "silo_engine_set_sharding: bad arguments");
   }
   engine->database.shard_rank = rank;
   engine->database.shard_world = world;
   engine->database.shard_by_position = shard_by_position != 0;
   engine->database.all_reduce = all_reduce;
   engine->database.all_reduce_context = context;
   return 0;
}

namespace {
// the native collectives of include/silo_gpu.h behind the engine's two hooks (context = the communicator)
int nativeAllReduce(void* context, uint32_t* device_values, size_t n, void* stream) {
   return silo_gpu_allreduce_counts(static_cast<silo_gpu_comm*>(context), device_values, n, stream);
}
int nativeBroadcast(void* context, void* device_bytes, size_t bytes, uint32_t root, void* stream) {
   return silo_gpu_broadcast_bytes(static_cast<silo_gpu_comm*>(context), device_bytes, bytes, root, stream);
}
}  // namespace

int silo_engine_set_comm(silo_engine* engine, silo_gpu_comm* comm, int shard_by_position) {
   if (engine == nullptr || comm == nullptr || silo_gpu_comm_world(comm) == 0) {
      return fail(SILO_GPU_ERR_INVALID_ARGUMENT, "silo_engine_set_comm: bad arguments");
   }
   engine->database.shard_rank = silo_gpu_comm_rank(comm);
   engine->database.shard_world = silo_gpu_comm_world(comm);
   engine->database.shard_by_position = shard_by_position != 0;
   engine->database.all_reduce = nativeAllReduce;
   engine->database.all_reduce_context = comm;
   engine->database.broadcast = shard_by_position != 0 ? nativeBroadcast : nullptr;
   engine->database.broadcast_context = shard_by_position != 0 ? comm : nullptr;
   return 0;
}

int silo_engine_set_broadcast(silo_engine* engine, silo_engine_broadcast_bytes broadcast, void* context) {
   if (engine == nullptr) {
      return fail(SILO_GPU_ERR_INVALID_ARGUMENT, "silo_engine_set_broadcast: null engine");
   }
   engine->database.broadcast = broadcast;
   engine->database.broadcast_context = context;
   return 0;
}

int silo_engine_set_option(silo_engine* engine, const char* name, int64_t value) {
   if (engine == nullptr || name == nullptr) {
      return fail(SILO_GPU_ERR_INVALID_ARGUMENT, "silo_engine_set_option: null argument");
   }
   if (std::strcmp(name, "mutation_row_capacity") == 0 && value >= 0 && value <= (1 << 24)) {
      engine->database.mutation_row_capacity = static_cast<uint32_t>(value);
      return 0;
   }
   if (std::strcmp(name, "compat_remove_quirk") == 0 && (value == 0 || value == 1)) {
      engine->database.compat_remove_quirk = value == 1;  // SILO_COMPAT_REMOVE_QUIRK, see filter_expressions.cpp
      return 0;
   }
   if (std::strcmp(name, "two_pass_build") == 0 && (value == 0 || value == 1)) {
      engine->database.two_pass_build = value == 1;
      return 0;
   }
   // How finalize lays THIS engine's stores out (silo_gpu_store_options; stores that are finalized already keep their layout):
   //   compact_scan_index  1 (default) = every position re-encoded into its cheapest layout, 0 = the 3 / 5 identity planes kept
   //   store_layout        the same choice in full: -1 identity planes, 0 cheapest layout with the most numerous symbol derived,
   //                       2 without one-hot rows, 3 with a one-hot row for the most numerous symbol too
   //   missing_symbol_runs 1 (default) = the missing symbol (N / X) kept as runs, 0 = as a plane
   if (std::strcmp(name, "compact_scan_index") == 0 && (value == 0 || value == 1)) {
      engine->database.store_options.layout = value == 1 ? 0 : -1;
      return guarded([&] { engine->database.applyStoreOptions(); return 0; });
   }
   if (std::strcmp(name, "store_layout") == 0 && (value == -1 || value == 0 || value == 2 || value == 3)) {
      engine->database.store_options.layout = static_cast<int32_t>(value);
      return guarded([&] { engine->database.applyStoreOptions(); return 0; });
   }
   if (std::strcmp(name, "missing_symbol_runs") == 0 && (value == 0 || value == 1)) {
      engine->database.store_options.missing_runs = value == 1 ? 0 : -1;
      return guarded([&] { engine->database.applyStoreOptions(); return 0; });
   }
   return fail(SILO_GPU_ERR_INVALID_ARGUMENT, std::string("silo_engine_set_option: unknown option or bad value: ") + name);
}

int silo_engine_execute_query(const silo_engine* engine, const char* query_json, char** out_json, int* out_http_status) {
   if (engine == nullptr || query_json == nullptr || out_json == nullptr || out_http_status == nullptr) {
      return fail(SILO_GPU_ERR_INVALID_ARGUMENT, "silo_engine_execute_query: null argument");
   }
   // the exception -> HTTP mapping of src/silo_api/query_handler.cpp:42-73
   try {
      *out_json = duplicate(engine->database.executeQueryJson(query_json));
      silo::Trace::mark("serialized");
      *out_http_status = 200;
   } catch (const silo::QueryParseException& ex) {
      *out_json = duplicate(errorDocument("Bad request", ex.what()));
      *out_http_status = 400;
   } catch (const std::exception& ex) {
      *out_json = duplicate(errorDocument("Internal Server Error", ex.what()));
      *out_http_status = 500;
   } catch (...) {
      *out_json = duplicate(errorDocument("Internal Server Error", "non recoverable error message"));
      *out_http_status = 500;
   }
   return *out_json != nullptr ? 0 : fail(SILO_GPU_ERR_OUT_OF_MEMORY, "out of memory");
}

int silo_engine_run_clients(
   const silo_engine* engine, const char* query_json, uint32_t n_clients, double seconds, uint64_t* out_queries, double* out_seconds,
   char** out_response
) {
   if (engine == nullptr || query_json == nullptr || n_clients == 0 || n_clients > 256 || out_queries == nullptr || out_seconds == nullptr) {
      return fail(SILO_GPU_ERR_INVALID_ARGUMENT, "silo_engine_run_clients: bad arguments");
   }
   std::atomic<uint64_t> answered{0};
   std::atomic<bool> failed{false};
   std::string last_response;
   const auto begin = std::chrono::steady_clock::now();
   const auto end = begin + std::chrono::duration_cast<std::chrono::steady_clock::duration>(std::chrono::duration<double>(seconds));
   std::vector<std::thread> clients;
   clients.reserve(n_clients);
   for (uint32_t client = 0; client < n_clients; ++client) {
      clients.emplace_back([&, client] {
         uint64_t mine = 0;
         while (!failed.load(std::memory_order_relaxed) && std::chrono::steady_clock::now() < end) {
            char* response = nullptr;
            int status = 0;
            if (silo_engine_execute_query(engine, query_json, &response, &status) != 0 || status != 200) {
               failed.store(true);
            } else {
               ++mine;
            }
            if (client == 0 && response != nullptr) {
               last_response = response;
            }
            free(response);
         }
         answered.fetch_add(mine);
      });
   }
   for (std::thread& client : clients) {
      client.join();
   }
   *out_seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - begin).count();
   *out_queries = answered.load();
   if (out_response != nullptr) {
      *out_response = duplicate(last_response);
   }
   if (failed.load()) {
      return fail(SILO_GPU_ERR_INVALID_ARGUMENT, "silo_engine_run_clients: a query was not answered with status 200: " + last_response);
   }
   return 0;
}

int silo_engine_evaluate_filter(
   const silo_engine* engine, const char* filter_json, int partition, uint64_t* out_bitset, size_t n_words, uint32_t* out_count, char** out_error_json,
   int* out_http_status
) {
   if (engine == nullptr || filter_json == nullptr || out_http_status == nullptr || partition < 0 ||
       static_cast<size_t>(partition) >= engine->database.partitions.size()) {
      return fail(SILO_GPU_ERR_INVALID_ARGUMENT, "silo_engine_evaluate_filter: bad arguments");
   }
   const silo::DatabasePartition& database_partition = engine->database.partitions[static_cast<size_t>(partition)];
   const size_t needed_words = (static_cast<size_t>(database_partition.sequence_count) + 63) / 64;
   if (out_bitset != nullptr && n_words < needed_words) {
      return fail(SILO_GPU_ERR_INVALID_ARGUMENT, "silo_engine_evaluate_filter: the bitset buffer is shorter than ceil(sequence_count / 64) words");
   }
   if (out_error_json != nullptr) {
      *out_error_json = nullptr;
   }
   const auto report = [&](const char* error, const std::string& message, int status) {
      if (out_error_json != nullptr) {
         *out_error_json = duplicate(errorDocument(error, message));
      }
      *out_http_status = status;
   };
   try {
      silo::checkGpu(silo_gpu_set_device(engine->database.device), "silo_gpu_set_device");
      silo::json::Value json;
      try {
         json = silo::json::parse(filter_json);
      } catch (const silo::json::ParseError& ex) {
         throw silo::QueryParseException("The query was not a valid JSON: " + std::string(ex.what()));
      }
      std::unique_ptr<silo::query_engine::filter_expressions::Expression> expression;
      try {
         expression = silo::query_engine::filter_expressions::parseExpression(json);
      } catch (const std::out_of_range& ex) {
         throw silo::QueryParseException("The query was not a valid JSON: " + std::string(ex.what()));
      }
      // Expression::compile + Operator::evaluate for this partition, query_engine.cpp:40-49
      const silo::query_engine::OperatorResult result = silo::query_engine::operators::Operator::evaluate(
         expression->compile(engine->database, database_partition, silo::query_engine::filter_expressions::Expression::AmbiguityMode::NONE)
      );
      if (out_bitset != nullptr) {
         std::memset(out_bitset, 0, n_words * sizeof(uint64_t));
         silo::checkGpu(
            silo_gpu_bitset_download(database_partition.store, out_bitset, result.bitset(), needed_words, silo::queryStream()), "silo_gpu_bitset_download"
         );
      }
      if (out_count != nullptr) {
         *out_count = result.cardinality();
      }
      *out_http_status = 200;
   } catch (const silo::QueryParseException& ex) {
      report("Bad request", ex.what(), 400);
   } catch (const std::exception& ex) {
      report("Internal Server Error", ex.what(), 500);
   } catch (...) {
      report("Internal Server Error", "non recoverable error message", 500);
   }
   return 0;
}

int silo_engine_execute_batch(const silo_engine* engine, const char* const* query_jsons, uint32_t n_queries, char** out_jsons, int* out_http_statuses) {
   if (engine == nullptr || (n_queries != 0 && (query_jsons == nullptr || out_jsons == nullptr || out_http_statuses == nullptr))) {
      return fail(SILO_GPU_ERR_INVALID_ARGUMENT, "silo_engine_execute_batch: null argument");
   }
   std::vector<std::string> queries;
   queries.reserve(n_queries);
   for (uint32_t i = 0; i < n_queries; ++i) {
      if (query_jsons[i] == nullptr) {
         return fail(SILO_GPU_ERR_INVALID_ARGUMENT, "silo_engine_execute_batch: null query");
      }
      queries.emplace_back(query_jsons[i]);
      out_jsons[i] = nullptr;
   }
   std::vector<silo::query_engine::QueryEngine::BatchOutcome> outcomes;
   std::exception_ptr whole_batch_error;
   try {
      const silo::query_engine::QueryEngine query_engine(engine->database);
      outcomes = query_engine.executeQueries(queries, true);
   } catch (...) {  // a failed shared launch fails every query of the batch
      whole_batch_error = std::current_exception();
      outcomes.assign(n_queries, {});
   }
   bool out_of_memory = false;
   for (uint32_t i = 0; i < n_queries; ++i) {
      // per query the exception -> HTTP mapping of src/silo_api/query_handler.cpp:42-73
      try {
         const std::exception_ptr error = whole_batch_error != nullptr ? whole_batch_error : outcomes[i].error;
         if (error != nullptr) {
            std::rethrow_exception(error);
         }
         out_jsons[i] = duplicate(outcomes[i].json);
         out_http_statuses[i] = 200;
      } catch (const silo::QueryParseException& ex) {
         out_jsons[i] = duplicate(errorDocument("Bad request", ex.what()));
         out_http_statuses[i] = 400;
      } catch (const std::exception& ex) {
         out_jsons[i] = duplicate(errorDocument("Internal Server Error", ex.what()));
         out_http_statuses[i] = 500;
      } catch (...) {
         out_jsons[i] = duplicate(errorDocument("Internal Server Error", "non recoverable error message"));
         out_http_statuses[i] = 500;
      }
      out_of_memory = out_of_memory || out_jsons[i] == nullptr;
   }
   if (out_of_memory) {
      for (uint32_t i = 0; i < n_queries; ++i) {
         std::free(out_jsons[i]);
         out_jsons[i] = nullptr;
      }
      return fail(SILO_GPU_ERR_OUT_OF_MEMORY, "out of memory");
   }
   return 0;
}

void silo_engine_free_string(char* text) {
   std::free(text);
}

void silo_engine_last_timings(int64_t* filter_microseconds, int64_t* action_microseconds) {
   const auto& timings = silo::Database::lastTimings();
   if (filter_microseconds != nullptr) {
      *filter_microseconds = timings.filter_microseconds;
   }
   if (action_microseconds != nullptr) {
      *action_microseconds = timings.action_microseconds;
   }
}

int silo_engine_data_version(const silo_engine* engine, char** out_text) {
   if (engine == nullptr || out_text == nullptr) {
      return fail(SILO_GPU_ERR_INVALID_ARGUMENT, "silo_engine_data_version: null argument");
   }
   *out_text = duplicate(engine->database.data_version);
   return *out_text != nullptr ? 0 : fail(SILO_GPU_ERR_OUT_OF_MEMORY, "out of memory");
}

int silo_engine_last_trace(char** out_json) {
   if (out_json == nullptr) {
      return fail(SILO_GPU_ERR_INVALID_ARGUMENT, "silo_engine_last_trace: null argument");
   }
   *out_json = duplicate(silo::Trace::json());
   return *out_json != nullptr ? 0 : fail(SILO_GPU_ERR_OUT_OF_MEMORY, "out of memory");
}

silo_gpu_store* silo_engine_partition_store(const silo_engine* engine, int partition) {
   if (engine == nullptr || partition < 0 || static_cast<size_t>(partition) >= engine->database.partitions.size()) {
      return nullptr;
   }
   return engine->database.partitions[static_cast<size_t>(partition)].store;
}

int silo_engine_position_window(const silo_engine* engine, const char* sequence_name, int is_amino_acid, uint32_t* begin, uint32_t* end) {
   if (engine == nullptr || sequence_name == nullptr || begin == nullptr || end == nullptr) {
      return fail(SILO_GPU_ERR_INVALID_ARGUMENT, "silo_engine_position_window: null argument");
   }
   size_t length = 0;
   if (is_amino_acid != 0) {
      const auto found = engine->database.aa_sequences.find(sequence_name);
      if (found == engine->database.aa_sequences.end()) {
         return fail(SILO_GPU_ERR_INVALID_ARGUMENT, "no such sequence store");
      }
      length = found->second.reference_sequence.size();
   } else {
      const auto found = engine->database.nuc_sequences.find(sequence_name);
      if (found == engine->database.nuc_sequences.end()) {
         return fail(SILO_GPU_ERR_INVALID_ARGUMENT, "no such sequence store");
      }
      length = found->second.reference_sequence.size();
   }
   const auto [window_begin, window_end] = engine->database.positionWindow(length);
   *begin = window_begin;
   *end = window_end;
   return 0;
}

int silo_engine_seqstore_id(const silo_engine* engine, int partition, const char* sequence_name, int is_amino_acid) {
   if (engine == nullptr || partition < 0 || static_cast<size_t>(partition) >= engine->database.partitions.size()) {
      return -1;
   }
   return seqstoreId(engine->database.partitions[static_cast<size_t>(partition)], sequence_name, is_amino_acid);
}

}  // extern "C"
