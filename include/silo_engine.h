/*
 * silo_engine.h — C ABI of the C++ host side (libsilo_engine.so): the QueryEngine-shaped layer that a
 * patched silo::Database::executeQuery (reference src/silo/database.cpp:710-714) would instantiate,
 * exported with plain C types so that non-C++ hosts (the Python tests / bench here) can drive it.
 *
 * One engine = one silo::Database whose partitions live on one GPU.  JSON in, JSON out:
 *   query   : the body of POST /query       (reference src/silo_api/query_handler.cpp:22-41)
 *   result  : {"queryResult":[...]}         (reference src/silo/query_engine/query_result.cpp:10-25)
 *   errors  : {"error":"Bad request"|"Internal Server Error","message":...} with HTTP status 400 / 500
 *             (reference src/silo_api/query_handler.cpp:42-73)
 */
#ifndef SILO_ENGINE_H
#define SILO_ENGINE_H

#include <stddef.h>
#include <stdint.h>

#include "silo_gpu.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct silo_engine silo_engine;

/* reference_genomes_json: contents of reference_genomes.json (reference_genomes.cpp);
 * alias_json: contents of pangolineage_alias.json or NULL (pango_lineage_alias.cpp:88-102);
 * default_nucleotide_sequence: NULL = "main" (database_config.cpp:70-75). */
int silo_engine_create(
   const char* reference_genomes_json, const char* alias_json, const char* default_nucleotide_sequence, int device, silo_engine** out
);
void silo_engine_destroy(silo_engine* engine);

/* Loads a data set directory in the reference's INPUT formats (preprocessing_config.yaml, database_config.yaml,
 * reference_genomes.json, pangolineage_alias.json, metadata TSV + FASTA[.zst|.xz] or ndjson[.zst|.xz]) into one
 * partition, rows in file order, and finalises it: the load side of Preprocessor::preprocess
 * (preprocessor.cpp:36-85) for this path, without DuckDB.  *out_summary_json (optional, malloc'ed) reports
 * {"sequenceCount":..,"nucleotideStores":..,"aminoAcidStores":..,"lineageColumns":..,"nullSequences":..}. */
int silo_engine_create_from_directory(const char* directory, int device, silo_engine** out, char** out_summary_json);

/* Adds a DatabasePartition of `sequence_count` rows; returns its index (>= 0) or a negative status. */
int silo_engine_add_partition(silo_engine* engine, uint32_t sequence_count);

/* Aligned sequences of one sequence store (nucleotide segment when is_amino_acid == 0, else gene),
 * row-major chars [n_sequences][length]; see silo_gpu_store_append_sequences. */
int silo_engine_append_sequences(
   silo_engine* engine, int partition, const char* sequence_name, int is_amino_acid, uint32_t first_sequence, uint32_t n_sequences,
   const char* chars, const uint8_t* is_null
);
int silo_engine_generate_synthetic(
   silo_engine* engine, int partition, const char* sequence_name, int is_amino_acid, const silo_gpu_synth_desc* synth
);
/* Two-pass build of one sequence store (silo_gpu_store_build_pass): pass 1 before its first sequences — the appends that follow
 * are only counted —, pass 2 before the same appends are repeated — they are written straight into the store's adaptive planes.
 * With option "two_pass_build" = 1 silo_engine_generate_synthetic does both passes by itself. */
int silo_engine_build_pass(silo_engine* engine, int partition, const char* sequence_name, int is_amino_acid, int pass);

/* A pango lineage metadata column: one raw (possibly aliased) value per row, NULL or "" = null. */
int silo_engine_set_lineage_column(silo_engine* engine, int partition, const char* column, const char* const* values, uint32_t n_rows);
/* Bulk form: dictionary of unaliased lineage names + one dictionary index per row. */
int silo_engine_set_lineage_column_ids(
   silo_engine* engine, int partition, const char* column, const char* const* dictionary, uint32_t n_dictionary, const uint32_t* value_ids,
   uint32_t n_rows
);

/* Call once after loading, before the first query. */
/* Metadata columns (the reference fills them from the metadata table in
 * Preprocessor::buildDatabase, preprocessor.cpp:447-503).  silo_engine_set_schema names the primary key and the
 * dateToSortBy column of database_config.yaml (call it before the first silo_engine_append_metadata).
 * silo_engine_append_metadata appends n_values rows in text form (NULL or "" = null value) to a column of a
 * partition; column_type is the type of database_config.yaml — "string", "indexed_string" (string with
 * generateIndex), "pango_lineage", "date", "int", "float", "insertion", "aaInsertion" — the column is created, and
 * appended to the schema, on its first call.  Every column must end up with one value per row of the partition.
 * A "pango_lineage" column also answers PangoLineage filters (no separate silo_engine_set_lineage_column needed). */
int silo_engine_set_schema(silo_engine* engine, const char* primary_key, const char* date_to_sort_by);
int silo_engine_append_metadata(
   silo_engine* engine, int partition, const char* column, const char* column_type, const char* const* values, uint32_t n_values
);
/* Unaligned nucleotide sequences for the Fasta action (fasta.cpp; the reference keeps them in zstd files next to the
 * database, unaligned_sequence_store.h): n_sequences strings (NULL = none) appended to a nucleotide sequence of a
 * partition.  Optional — rows without one answer null.  They stay in host memory. */
int silo_engine_append_unaligned_sequences(
   silo_engine* engine, int partition, const char* sequence_name, const char* const* sequences, uint32_t n_sequences
);
int silo_engine_finalize(silo_engine* engine);

/* Multi-GPU (one process per GPU).  Call before silo_engine_add_partition.  shard_by_position != 0:
 * this rank holds and scans only positions [P*rank/world, P*(rank+1)/world) of every sequence store
 * (sequences appended must be that slice) and counts are all-reduced; otherwise the partitions of
 * this rank are a sequence-id shard and counts / cardinalities are all-reduced.
 *
 * silo_engine_set_comm is the production form: rank and world are the communicator's, the count tables are summed by
 * silo_gpu_allreduce_counts (ncclAllReduce over xGMI) and — under position sharding — filter leaves travel by
 * silo_gpu_broadcast_bytes, both enqueued on the HIP stream of the request thread that runs the query (per-thread
 * query streams stay in use; no host synchronisation around the collective).  The communicator must outlive the
 * engine.  Every rank has to run the same queries in the same order (SPMD): the collectives of concurrent request
 * threads are serialised per communicator, but their order across ranks is the caller's to keep.  Every all-reduce of a
 * query carries the fingerprint of the query text; ranks that ran different queries answer 500 ("the ranks of this sharded
 * database did not run the same query") instead of mixing their counts.
 *
 * silo_engine_set_sharding / silo_engine_set_broadcast install caller-supplied collectives instead (tests back them
 * with gloo through host memory; a host with its own transport can plug it in): all_reduce sums n uint32 in place on
 * the device across ranks, and is handed the stream the engine's kernels before and after it run on. */
int silo_engine_set_comm(silo_engine* engine, silo_gpu_comm* comm, int shard_by_position);
typedef int (*silo_engine_all_reduce_u32)(void* context, uint32_t* device_values, size_t n, void* stream);
int silo_engine_set_sharding(
   silo_engine* engine, uint32_t rank, uint32_t world, int shard_by_position, silo_engine_all_reduce_u32 all_reduce, void* context
);

/* Position-range sharding only: broadcast `bytes` device bytes in place from rank `root` to all ranks.
 * With it installed, a filter leaf at a position another rank owns is fetched from that rank (every
 * rank must run the same queries in the same order); without it such a leaf is a 500 "not resident". */
typedef int (*silo_engine_broadcast_bytes)(void* context, void* device_bytes, size_t bytes, uint32_t root, void* stream);
int silo_engine_set_broadcast(silo_engine* engine, silo_engine_broadcast_bytes broadcast, void* context);

/* Tunables.  "mutation_row_capacity": Mutations / AminoAcidMutations select their result rows on the device into a
 * list of this many cells per query (default 4096); a query selecting more fetches the whole count table
 * and selects on the host; 0 = always the host selection.  Results are identical either way.
 * How silo_engine_finalize lays THIS engine's sequence stores out (per engine: silo_gpu_store_options of its device stores; set
 * before finalize, finalized stores keep their layout):
 *   "compact_scan_index" (1 default / 0): every position re-encoded into its cheapest layout — at almost every position of an
 *     alignment the most numerous symbol is stored nowhere and derived by the scan, the rest are one-hot rows and keys — or (0) the
 *     3 / 5 identity code planes kept as built (every cell read by a scan: what a query costs when no position has a dominant symbol);
 *   "store_layout" (-1 / 0 default / 2 / 3): the same choice in full, as SILO_GPU_TUNE_COMPACT_INDEX of include/silo_gpu.h;
 *   "missing_symbol_runs" (1 default / 0): the missing symbol (N / X) kept as runs along the rows, or as a plane per position
 *     (then no symbol is derived).
 * "two_pass_build" (0 default / 1): silo_engine_generate_synthetic runs the generator twice per sequence store (counted, then written
 * straight into the finished layout: no build-time planes); chosen by itself where the one-pass build would not fit the device.
 * "compat_remove_quirk" (1 default / 0): SILO_COMPAT_REMOVE_QUIRK — HasNucleotideMutation / HasAminoAcidMutation build their
 * symbol lists as the reference does, with std::remove and no erase (has_mutation.cpp:58-65, has_aa_mutation.cpp:48-52):
 * at a reference-T (amino acids: reference-STOP) position the reference symbol itself stays in the list.  0 = the list
 * without the reference symbol, which is what that code meant.  Unknown name: error. */
int silo_engine_set_option(silo_engine* engine, const char* name, int64_t value);

/* Executes one query.  *out_json is malloc'ed (free with silo_engine_free_string) and holds either the
 * result or the error document; *out_http_status is 200, 400 or 500.  Returns 0 unless the arguments
 * themselves are invalid.  Re-entrant: may be called from many threads on one engine. */
int silo_engine_execute_query(const silo_engine* engine, const char* query_json, char** out_json, int* out_http_status);

/* Measurement helper (bench.py, tools/): `n_clients` request threads — what silo_api's request handler threads are to the
 * reference — each call silo_engine_execute_query with the same query, one query at a time, for `seconds`; *out_queries =
 * queries answered with status 200 by all of them, *out_seconds = the wall time they took, *out_response (malloc'ed, may be
 * NULL to ignore) = the last response of client 0.  Any other status stops the run and is returned as an error.  The clients
 * are native threads: a Python caller's interpreter lock is not part of the figure. */
int silo_engine_run_clients(
   const silo_engine* engine, const char* query_json, uint32_t n_clients, double seconds, uint64_t* out_queries, double* out_seconds,
   char** out_response
);

/* The inner seam of SURVEY.md §8(b): Expression::compile + Operator::evaluate of ONE filter for ONE partition
 * (query_engine.cpp:40-49; operator.h:32 `evaluate() -> OperatorResult`), handing back what the reference's
 * OperatorResult holds — the set of sequence ids — as a bitset in host memory (bit i of word w = row 64 * w + i of the
 * partition; n_words >= ceil(sequence_count / 64), the rest is zeroed; may be NULL) and its cardinality (may be NULL).
 * filter_json is a filterExpression object.  *out_http_status is 200, or 400 / 500 with the error document in
 * *out_error_json (malloc'ed, may be NULL to ignore) exactly as silo_engine_execute_query would report it. */
int silo_engine_evaluate_filter(
   const silo_engine* engine, const char* filter_json, int partition, uint64_t* out_bitset, size_t n_words, uint32_t* out_count,
   char** out_error_json, int* out_http_status
);

/* Executes `n_queries` queries as one batch: every query is parsed, compiled and its filter evaluated, then
 * the Mutations / AminoAcidMutations scans of all of them are launched together so that queries over the same
 * sequence store share passes over the planes (up to SILO_GPU_MAX_SCAN_BATCH filters per pass), then the rows
 * of each query are built.  Results are exactly those of n_queries silo_engine_execute_query calls, in order:
 * out_jsons[i] (malloc'ed, free each with silo_engine_free_string) and out_http_statuses[i] per query; one
 * failing query does not affect the others.  Stands where silo_api's request handler (src/silo_api/
 * query_handler.cpp:26-73) would hand several queued requests to the engine at once.  Returns 0 unless the
 * arguments themselves are invalid.  With collectives installed (silo_engine_set_sharding) every rank must pass the
 * same batch; the count tables of the batch are all-reduced after its scans were launched, in query order. */
int silo_engine_execute_batch(
   const silo_engine* engine, const char* const* query_jsons, uint32_t n_queries, char** out_jsons, int* out_http_statuses
);
void silo_engine_free_string(char* text);

/* The reference's two per-query timings (query_engine.cpp:63-65) of the last query on this thread. */
void silo_engine_last_timings(int64_t* filter_microseconds, int64_t* action_microseconds);

/* The value of the `data-version` response header (query_handler.cpp:38, data_version.cpp:9-13): the unix time at which
 * silo_engine_finalize built the data, as decimal text; *out_text is malloc'ed (free with silo_engine_free_string). */
int silo_engine_data_version(const silo_engine* engine, char** out_text);

/* Phase marks of the last query on this thread as JSON {"phase": microseconds since the query began, ...};
 * *out_json is malloc'ed (free with silo_engine_free_string). */
int silo_engine_last_trace(char** out_json);

/* Device store of a partition, for callers that drive the kernels directly (bench roofline leg). */
silo_gpu_store* silo_engine_partition_store(const silo_engine* engine, int partition);
/* silo_gpu sequence-store index of a named store inside a partition, or -1. */
int silo_engine_seqstore_id(const silo_engine* engine, int partition, const char* sequence_name, int is_amino_acid);

/* Genome positions [*begin, *end) of a named sequence store that are resident on this rank. */
int silo_engine_position_window(const silo_engine* engine, const char* sequence_name, int is_amino_acid, uint32_t* begin, uint32_t* end);

const char* silo_engine_last_error(void);

#ifdef __cplusplus
}
#endif
#endif /* SILO_ENGINE_H */
