// database.h — host-side mirror of the slice of silo::Database the query path reads.
//
// Reference types mirrored (file:line under the reference tree):
//   silo::Database                       include/silo/database.h:29-118 (partitions, nuc_sequences, aa_sequences,
//                                        alias_key, database_config, executeQuery :79)
//   silo::DatabasePartition              include/silo/storage/database_partition.h:39-112
//   silo::SequenceStorePartition<S>      include/silo/storage/sequence_store.h:34-88
//   silo::SequenceStore<S>               include/silo/storage/sequence_store.h:90-101
//   storage::column::PangoLineageColumnPartition   include/silo/storage/column/pango_lineage_column.h
//   silo::PangoLineageAliasLookup        include/silo/storage/pango_lineage_alias.h
//
// MI355X-first difference: the bitmaps live in HBM behind a silo_gpu_store (include/silo_gpu.h); this
// layer holds names, reference sequences, the lineage dictionary and device handles only.
#pragma once

#include <cstdint>
#include <deque>
#include <functional>
#include <map>
#include <memory>
#include <mutex>
#include <shared_mutex>
#include <optional>
#include <string>
#include <unordered_map>
#include <vector>

#include "json.h"
#include "silo_gpu.h"
#include "metadata_columns.h"
#include "symbols.h"

namespace silo {

namespace query_engine {
struct QueryResult;
}

/// Thrown when a silo_gpu call fails; surfaces as HTTP 500 like any std::exception
/// (src/silo_api/query_handler.cpp:46-50).
class DeviceException : public std::runtime_error {
  public:
   using std::runtime_error::runtime_error;
};
void checkGpu(int status, const char* what);

/// The HIP stream the calling thread's query work is issued on.  Every request thread gets its own
/// non-blocking stream, so concurrent executeQuery calls (silo_api runs one thread per request,
/// database_mutex.cpp:20-23) overlap on the device instead of serialising on the null stream.  Collectives of a
/// sharded database (the all-reduce of a count table, the broadcast of a filter leaf) are enqueued on this same
/// stream — natively through silo_gpu_allreduce_counts / silo_gpu_broadcast_bytes, or by a caller-supplied
/// callback that is handed the stream — so sharding does not cost the per-request concurrency.
void* queryStream();
/// The device of this process's engine (one process per GPU): what a new host thread selects before it creates its stream.
void setEngineDevice(int device);

/// Phase marks of the calling thread's last query, microseconds since the query began — the finer
/// grained companion of the reference's two LOG_PERFORMANCE timings (query_engine.cpp:63-65).
struct Trace {
   static void reset();
   static void mark(const char* name);
   static std::string json();  // {"phase": microseconds, ...} in mark order
};

/// RAII device allocation taken from / returned to a per-database pool (no hipMalloc on the hot path).
class DevicePool;
class DeviceBuffer {
  public:
   DeviceBuffer() = default;
   DeviceBuffer(DevicePool* pool, void* ptr, size_t bytes) : pool_(pool), ptr_(ptr), bytes_(bytes) {}
   DeviceBuffer(DeviceBuffer&& other) noexcept { *this = std::move(other); }
   DeviceBuffer& operator=(DeviceBuffer&& other) noexcept;
   DeviceBuffer(const DeviceBuffer&) = delete;
   DeviceBuffer& operator=(const DeviceBuffer&) = delete;
   ~DeviceBuffer();
   [[nodiscard]] void* get() const { return ptr_; }
   template <typename T>
   [[nodiscard]] T* as() const { return static_cast<T*>(ptr_); }
   [[nodiscard]] size_t size() const { return bytes_; }
   explicit operator bool() const { return ptr_ != nullptr; }

  private:
   DevicePool* pool_ = nullptr;
   void* ptr_ = nullptr;
   size_t bytes_ = 0;
};

class DevicePool {
  public:
   ~DevicePool();
   DeviceBuffer acquire(size_t bytes);
   void release(void* ptr, size_t bytes);

  private:
   std::mutex mutex_;
   std::multimap<size_t, void*> free_;
};

/// A device -> host transfer in flight: a page-locked host buffer and the event recorded after the copy.
/// Buffers and events are recycled through a process-wide free list (hipHostMalloc / hipEventCreate are slow).
class HostFetch {
  public:
   HostFetch() = default;
   /// Enqueues the copy of `bytes` from `device_source` on `stream` and records the completion event.
   HostFetch(const void* device_source, size_t bytes, void* stream);
   HostFetch(HostFetch&& other) noexcept { *this = std::move(other); }
   HostFetch& operator=(HostFetch&& other) noexcept;
   HostFetch(const HostFetch&) = delete;
   HostFetch& operator=(const HostFetch&) = delete;
   ~HostFetch();
   /// Blocks until the copy has landed; the data stays valid as long as this object lives.
   [[nodiscard]] const void* wait() const;
   explicit operator bool() const { return host_ != nullptr; }

  private:
   void* host_ = nullptr;
   void* event_ = nullptr;
   size_t capacity_ = 0;
};

/// A row slot of the device library (silo_gpu_row_slot: the selected rows of a Mutations query written by the kernel straight
/// into page-locked host memory) out of a process-wide free list; goes back when the object dies, after its launch has
/// delivered.
class RowSlot {
  public:
   RowSlot() = default;
   explicit RowSlot(uint32_t row_capacity);
   RowSlot(RowSlot&& other) noexcept { *this = std::move(other); }
   RowSlot& operator=(RowSlot&& other) noexcept;
   RowSlot(const RowSlot&) = delete;
   RowSlot& operator=(const RowSlot&) = delete;
   ~RowSlot();
   /// k_mutations_select over `counts` with the rows delivered into this slot; enqueued on `stream`.
   void select(const uint32_t* counts, const uint8_t* reference_index, uint32_t n_positions, uint32_t n_symbols, double min_proportion, void* stream);
   /// Blocks (spins) until the launch has delivered: the rows and the number of selected cells (may exceed the capacity).
   std::pair<const silo_gpu_mutation_row*, uint32_t> wait();
   explicit operator bool() const { return slot_ != nullptr; }
   [[nodiscard]] uint32_t capacity() const { return capacity_; }

  private:
   silo_gpu_row_slot* slot_ = nullptr;
   uint32_t capacity_ = 0;
   void* stream_ = nullptr;
   bool in_flight_ = false;
};

class PangoLineageAliasLookup {
  public:
   PangoLineageAliasLookup() = default;
   explicit PangoLineageAliasLookup(std::unordered_map<std::string, std::vector<std::string>> alias_key)
       : alias_key(std::move(alias_key)) {}
   static PangoLineageAliasLookup fromJson(const json::Value& json);
   [[nodiscard]] std::string unaliasPangoLineage(const std::string& pango_lineage) const;
   [[nodiscard]] std::string aliasPangoLineage(const std::string& unaliased_pango_lineage) const;

  private:
   std::unordered_map<std::string, std::vector<std::string>> alias_key;
};

std::vector<std::string> getParentLineages(const std::string& unaliased_lineage);

class DatabasePartition;

namespace storage::column {

/// Dictionary-encoded lineage column: the value id of every row sits in HBM; a filter bitset is built on
/// the device from a per-dictionary-entry membership table on first use and cached ("host-built bitset
/// uploaded once").  Reference: pango_lineage_column.cpp:21-77.
class PangoLineageColumnPartition {
  public:
   PangoLineageColumnPartition(const PangoLineageAliasLookup& alias_key, const DatabasePartition& partition);
   ~PangoLineageColumnPartition();
   PangoLineageColumnPartition(const PangoLineageColumnPartition&) = delete;

   void insert(const std::string& value);
   void insertNull() { insert(""); }
   /// Bulk form for synthetic data: ids into `dictionary` (already unaliased lineage names).
   void setValues(std::vector<std::string> dictionary, const uint32_t* value_ids, size_t n_rows);
   void finalize();

   [[nodiscard]] std::optional<const uint64_t*> filter(const std::string& value) const;
   [[nodiscard]] std::optional<const uint64_t*> filterIncludingSublineages(const std::string& value) const;
   [[nodiscard]] size_t numRows() const { return n_rows_; }

  private:
   std::optional<const uint64_t*> lookup(const std::string& value, bool sublineages) const;

   const PangoLineageAliasLookup& alias_key;
   const DatabasePartition& partition;
   std::vector<std::string> dictionary_;  // unaliased lineage per value id
   std::unordered_map<std::string, uint32_t> lookup_unaliased_;
   std::vector<uint32_t> value_ids_;      // host staging until finalize()
   uint32_t* d_value_ids_ = nullptr;
   size_t n_rows_ = 0;
   mutable std::mutex mutex_;
   mutable std::map<std::pair<std::string, bool>, uint64_t*> cache_;  // device bitsets
};

}  // namespace storage::column

template <typename SymbolType>
class SequenceStore {
  public:
   std::vector<typename SymbolType::Symbol> reference_sequence;
};

/// Layout of the count table of a Mutations query: the stores of one alphabet back to back in name order,
/// counts[(offset(store) + position) * n_valid_symbols + symbol].  One table per query means one memset, one
/// all-reduce, one row selection (k_mutations_select) and one transfer however many stores are scanned.
struct MutationTableLayout {
   std::map<std::string, uint32_t> position_offset;
   uint32_t total_positions = 0;
   /// Per table position the index of the reference symbol within VALID_MUTATION_SYMBOLS (0xFF: not among them),
   /// on the device; uploaded by Database::finalize.
   std::shared_ptr<const uint8_t> reference_index_device;
};

template <typename SymbolType>
class SequenceStorePartition {
  public:
   SequenceStorePartition(
      const std::vector<typename SymbolType::Symbol>& reference_sequence, silo_gpu_store* store, uint32_t seqstore_id,
      uint32_t sequence_count, uint32_t position_begin, uint32_t position_end
   )
       : reference_sequence(reference_sequence),
         store(store),
         seqstore_id(seqstore_id),
         sequence_count(sequence_count),
         position_begin(position_begin),
         position_end(position_end) {}

   const std::vector<typename SymbolType::Symbol>& reference_sequence;
   silo_gpu_store* store;
   uint32_t seqstore_id;
   uint32_t sequence_count;
   /// Positions [position_begin, position_end) of the genome are resident in this rank's HBM
   /// (the whole genome unless the database is sharded by position range).
   uint32_t position_begin;
   uint32_t position_end;

   [[nodiscard]] bool holds(size_t position) const { return position >= position_begin && position < position_end; }

   /// Device pointer of the dense plane, nullptr when the symbol is stored sparsely
   /// (sequence_store.cpp:92-98 returned a roaring pointer).  `position` is a genome position.
   [[nodiscard]] const uint64_t* getBitmap(size_t position, typename SymbolType::Symbol symbol) const {
      if (!holds(position)) {
         return nullptr;
      }
      return silo_gpu_store_plane(store, seqstore_id, static_cast<uint32_t>(position - position_begin), static_cast<uint32_t>(symbol));
   }
};

class DatabasePartition {
  public:
   DatabasePartition() = default;
   DatabasePartition(const DatabasePartition&) = delete;
   ~DatabasePartition();

   uint32_t sequence_count = 0;
   silo_gpu_store* store = nullptr;
   std::map<std::string, SequenceStorePartition<Nucleotide>> nuc_sequences;
   std::map<std::string, SequenceStorePartition<AminoAcid>> aa_sequences;
   /// Unaligned nucleotide sequences by sequence name (unaligned_sequence_store.h), one optional string per row.  The
   /// reference keeps them zstd-compressed on disk and reads them back through DuckDB (fasta.cpp:60-211); they never
   /// touch the device, so here they simply stay on the host for the Fasta action.
   std::map<std::string, std::vector<std::optional<std::string>>> unaligned_nuc_sequences;
   struct ColumnPartitionGroup {
      std::map<std::string, storage::column::PangoLineageColumnPartition> pango_lineage_columns;
      /// Every metadata column of database_config.metadata by name (column_group.h keeps one map per type; the
      /// type is a member here).  A lineage column appears in both maps: the one above answers PangoLineage
      /// filters, this one renders and groups its values.
      std::map<std::string, storage::column::MetadataColumnPartition> metadata_columns;
      /// The insertion indexes behind InsertionContains / Insertions; the text of an insertion column is in
      /// metadata_columns like any other string column.
      std::map<std::string, storage::column::InsertionColumnPartition> nuc_insertion_columns;
      std::map<std::string, storage::column::InsertionColumnPartition> aa_insertion_columns;
      template <typename SymbolType>
      [[nodiscard]] const std::map<std::string, storage::column::InsertionColumnPartition>& getInsertionColumns() const;

      [[nodiscard]] const storage::column::MetadataColumnPartition* find(const std::string& name, config::ColumnType type) const {
         const auto found = metadata_columns.find(name);
         return found != metadata_columns.end() && found->second.type == type ? &found->second : nullptr;
      }
   } columns;

   mutable DevicePool pool;

   /// Materialised bitsets of sparsely stored symbols (IUPAC ambiguity codes), keyed by
   /// seqstore << 40 | local position << 8 | symbol; filled on first use by ProgramBuilder::sparseLeaf.
   static constexpr size_t SPARSE_CACHE_BYTES = size_t{32} << 30;
   /// readers (every filter leaf of every query looks its plane up here) share the lock; only the first use of a plane writes
   mutable std::shared_mutex sparse_cache_mutex;
   mutable std::map<uint64_t, DeviceBuffer> sparse_cache;
   /// Row bitsets of the values of INDEXED string columns, built on first use (the reference builds one roaring bitmap
   /// per value at insert time, indexed_string_column.cpp:24-36); same mutex, own budget.
   static constexpr size_t INDEXED_VALUE_CACHE_BYTES = size_t{1} << 30;
   mutable std::map<std::pair<const storage::column::MetadataColumnPartition*, uint32_t>, DeviceBuffer> indexed_value_cache;

   /// Cardinalities of stored row bitsets (lineage sets, missing-symbol planes, derived planes) by device address: stored
   /// bitmaps never change once the database is finalized, and roaring keeps a bitmap's cardinality at hand
   /// (mutations.cpp:45 reads it for free) — a filter that IS a stored bitmap costs no popcount launch and no copy here either.
   mutable std::shared_mutex cardinality_cache_mutex;
   mutable std::unordered_map<const uint64_t*, uint32_t> cardinality_cache;

   [[nodiscard]] uint32_t rowWords() const { return silo_gpu_store_row_words(store); }

   template <typename SymbolType>
   [[nodiscard]] const std::map<std::string, SequenceStorePartition<SymbolType>>& getSequenceStores() const;
};

/// Collective hook for multi-GPU runs (one process per GPU): sums `n` uint32 on the device across all
/// ranks in place, on `stream`.  bench.py backs it with torch.distributed (RCCL over xGMI); a native
/// host would call ncclAllReduce.  SURVEY.md §8(e).
using AllReduceU32 = int (*)(void* context, uint32_t* device_values, size_t n, void* stream);
/// Broadcast of `bytes` device bytes from rank `root` to every rank (in place).  Used under position-range
/// sharding to hand a filter leaf (one 1.25 MB plane at 10 M sequences) from the rank that owns its
/// position to the others; every rank runs the same queries in the same order (SPMD).
using BroadcastBytes = int (*)(void* context, void* device_bytes, size_t bytes, uint32_t root, void* stream);

class Database {
  public:
   Database() = default;
   virtual ~Database() = default;

   std::deque<DatabasePartition> partitions;
   std::map<std::string, SequenceStore<Nucleotide>> nuc_sequences;
   std::map<std::string, SequenceStore<AminoAcid>> aa_sequences;
   /// data_version.cpp:9-13: the moment the data was built, as a decimal unix time; silo_api sends it as the
   /// `data-version` header of every query response (query_handler.cpp:38).  Set by finalize().
   std::string data_version;
   /// database_config.yaml: default nucleotide sequence; metadata columns in file order, primary key, dateToSortBy
   config::DatabaseConfig database_config;
   PangoLineageAliasLookup alias_key;
   int device = 0;

   // --- sharding ---------------------------------------------------------------
   /// Position-range sharding: this rank holds and scans positions [P*rank/world, P*(rank+1)/world) of
   /// every sequence store and the counts are all-reduced (set before addPartition).  Otherwise the
   /// partitions of this rank are a sequence-id shard.  world == 1 -> everything local.
   uint32_t shard_rank = 0;
   uint32_t shard_world = 1;
   bool shard_by_position = false;
   /// Mutations selects its result rows on the device into a list of this many cells; a query that selects more
   /// (minProportion 0 over a large filter) fetches the whole count table instead. 0 = always fetch the table.
   uint32_t mutation_row_capacity = 4096;
   /// How the stores of THIS database are laid out at finalize (silo_gpu_store_options): a field left at
   /// SILO_GPU_OPTION_DEFAULT follows the process-wide probe knob (silo_gpu_tune).
   silo_gpu_store_options store_options{SILO_GPU_OPTION_DEFAULT, SILO_GPU_OPTION_DEFAULT, SILO_GPU_OPTION_DEFAULT, SILO_GPU_OPTION_DEFAULT};
   /// Hands store_options to every partition's device store (those not finalized yet take them up).
   void applyStoreOptions();
   /// Option "two_pass_build": generated (and directory-loaded) sequence stores are streamed twice — counted, then written
   /// straight into their adaptive planes — instead of being built in 3 / 5 identity planes per position and re-encoded.
   bool two_pass_build = false;
   /// SILO_COMPAT_REMOVE_QUIRK: HasNucleotideMutation / HasAminoAcidMutation keep the reference's
   /// std::remove-without-erase behaviour (filter_expressions.cpp, dropSymbol).  Default on: parity is judged against
   /// the reference as it is.
#ifndef SILO_COMPAT_REMOVE_QUIRK
#define SILO_COMPAT_REMOVE_QUIRK 1
#endif
   bool compat_remove_quirk = SILO_COMPAT_REMOVE_QUIRK != 0;
   AllReduceU32 all_reduce = nullptr;
   void* all_reduce_context = nullptr;
   BroadcastBytes broadcast = nullptr;
   void* broadcast_context = nullptr;

   /// Rank whose position window contains `position` of a genome of `length` positions.
   [[nodiscard]] uint32_t ownerOfPosition(size_t position, size_t length) const;

   /// Timings of the last query on this thread, the reference's two phases (query_engine.cpp:63-65).
   struct Timings {
      int64_t filter_microseconds = 0;
      int64_t action_microseconds = 0;
   };
   static Timings& lastTimings();
   /// Fingerprint (FNV-1a of the query text) of the query the calling thread is executing.  A sharded database adds it to
   /// every all-reduce of the query: ranks that did not run the same query see sums that are not `world` times their own
   /// fingerprint and answer 500 instead of mixing the counts of different queries (checkSameQuery in actions.cpp).
   static uint64_t& queryFingerprint();
   static uint64_t fingerprintOf(const std::string& query_text);

   template <typename SymbolType>
   [[nodiscard]] const std::map<std::string, SequenceStore<SymbolType>>& getSequenceStores() const;
   MutationTableLayout nuc_mutation_layout;
   MutationTableLayout aa_mutation_layout;
   template <typename SymbolType>
   [[nodiscard]] const MutationTableLayout& getMutationTableLayout() const;

   /// database.cpp:710-714
   [[nodiscard]] virtual query_engine::QueryResult executeQuery(const std::string& query) const;
   /// The same as the response body silo_api's QueryHandler::post would write (query_handler.cpp:38-41).
   [[nodiscard]] std::string executeQueryJson(const std::string& query) const;

   /// The position range [begin, end) of a genome of `length` positions that this rank holds.
   [[nodiscard]] std::pair<uint32_t, uint32_t> positionWindow(size_t length) const;

   // --- construction (replaces Preprocessor::buildDatabase, preprocessor.cpp:447-503) -----------
   void setReferenceGenomes(const json::Value& reference_genomes);
   DatabasePartition& addPartition(uint32_t sequence_count);
   /// Appends `values` (text form, "" = NULL) to metadata column `name` of `partition`, creating the column and its
   /// entry in database_config.metadata on first use.  A lineage column also feeds the PangoLineage filter index.
   /// Replaces the column inserts of Preprocessor::buildDatabase (preprocessor.cpp:447-503, column_group.cpp).
   void appendMetadata(DatabasePartition& partition, const std::string& name, config::ColumnType type, const std::vector<std::string>& values);
   /// Appends unaligned nucleotide sequences (nullopt = none) of `sequence_name` to `partition`.
   void appendUnalignedSequences(DatabasePartition& partition, const std::string& sequence_name, std::vector<std::optional<std::string>> values);
   void finalize();
};

template <>
inline const std::map<std::string, SequenceStorePartition<Nucleotide>>& DatabasePartition::getSequenceStores<Nucleotide>() const {
   return nuc_sequences;
}
template <>
inline const std::map<std::string, SequenceStorePartition<AminoAcid>>& DatabasePartition::getSequenceStores<AminoAcid>() const {
   return aa_sequences;
}
template <>
inline const std::map<std::string, SequenceStore<Nucleotide>>& Database::getSequenceStores<Nucleotide>() const {
   return nuc_sequences;
}
template <>
inline const std::map<std::string, SequenceStore<AminoAcid>>& Database::getSequenceStores<AminoAcid>() const {
   return aa_sequences;
}

template <>
inline const std::map<std::string, storage::column::InsertionColumnPartition>& DatabasePartition::ColumnPartitionGroup::getInsertionColumns<
   Nucleotide>() const {
   return nuc_insertion_columns;
}
template <>
inline const std::map<std::string, storage::column::InsertionColumnPartition>& DatabasePartition::ColumnPartitionGroup::getInsertionColumns<
   AminoAcid>() const {
   return aa_insertion_columns;
}
template <>
inline const MutationTableLayout& Database::getMutationTableLayout<Nucleotide>() const {
   return nuc_mutation_layout;
}
template <>
inline const MutationTableLayout& Database::getMutationTableLayout<AminoAcid>() const {
   return aa_mutation_layout;
}

}  // namespace silo
