#include "database.h"

#include <algorithm>
#include <atomic>
#include <chrono>
#include <sstream>

#include "query_engine.h"

namespace silo {

void checkGpu(int status, const char* what) {
   if (status != SILO_GPU_OK) {
      throw DeviceException(std::string(what) + ": " + silo_gpu_last_error());
   }
}

// ---- per-thread streams ---------------------------------------------------------------------------
namespace {

struct ThreadStream {  // never destroyed: thread exit may come after the HIP runtime has shut down
   void* stream = nullptr;
   bool tried = false;
};

}  // namespace

namespace {
std::atomic<int> g_engine_device{0};
}

void setEngineDevice(int device) {
   g_engine_device.store(device);
}

void* queryStream() {
   thread_local ThreadStream holder;
   if (!holder.tried) {
      holder.tried = true;
      // HIP's current device is per host thread and starts at 0: a request thread or batch worker of the process that serves
      // GPU r (one process per GPU, all devices visible) must create its stream — and everything else it creates — on r
      (void)silo_gpu_set_device(g_engine_device.load());
      if (silo_gpu_stream_create(&holder.stream) != SILO_GPU_OK) {
         holder.stream = nullptr;  // fall back to the null stream
      }
   }
   return holder.stream;
}

// ---- trace --------------------------------------------------------------------------------------
namespace {

struct TraceState {
   std::chrono::steady_clock::time_point start = std::chrono::steady_clock::now();
   std::vector<std::pair<const char*, int64_t>> marks;
};

TraceState& traceState() {
   thread_local TraceState state;
   return state;
}

}  // namespace

void Trace::reset() {
   TraceState& state = traceState();
   state.start = std::chrono::steady_clock::now();
   state.marks.clear();
}

void Trace::mark(const char* name) {
   TraceState& state = traceState();
   state.marks.emplace_back(
      name, std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now() - state.start).count()
   );
}

std::string Trace::json() {
   std::string out = "{";
   bool first = true;
   for (const auto& [name, microseconds] : traceState().marks) {
      out += std::string(first ? "" : ",") + "\"" + name + "\":" + std::to_string(microseconds);
      first = false;
   }
   return out + "}";
}

// ---- device pool --------------------------------------------------------------------------------
DeviceBuffer& DeviceBuffer::operator=(DeviceBuffer&& other) noexcept {
   if (this != &other) {
      if (ptr_ != nullptr && pool_ != nullptr) {
         pool_->release(ptr_, bytes_);
      }
      pool_ = other.pool_;
      ptr_ = other.ptr_;
      bytes_ = other.bytes_;
      other.ptr_ = nullptr;
      other.pool_ = nullptr;
      other.bytes_ = 0;
   }
   return *this;
}

DeviceBuffer::~DeviceBuffer() {
   if (ptr_ != nullptr && pool_ != nullptr) {
      pool_->release(ptr_, bytes_);
   }
}

DevicePool::~DevicePool() {
   for (auto& [bytes, ptr] : free_) {
      silo_gpu_free(ptr);
   }
}

DeviceBuffer DevicePool::acquire(size_t bytes) {
   bytes = std::max<size_t>(bytes, 256);
   {
      const std::lock_guard<std::mutex> lock(mutex_);
      const auto found = free_.find(bytes);
      if (found != free_.end()) {
         void* ptr = found->second;
         free_.erase(found);
         return {this, ptr, bytes};
      }
   }
   void* ptr = nullptr;
   checkGpu(silo_gpu_malloc(bytes, &ptr), "silo_gpu_malloc");
   return {this, ptr, bytes};
}

void DevicePool::release(void* ptr, size_t bytes) {
   const std::lock_guard<std::mutex> lock(mutex_);
   free_.emplace(bytes, ptr);
}

// ---- HostFetch ---------------------------------------------------------------------------------
namespace {
struct StagingSlot {
   void* host;
   void* event;
};
// never destroyed: threads may still return slots while the process is shutting down
std::mutex& stagingMutex() {
   static auto* mutex = new std::mutex;
   return *mutex;
}
std::multimap<size_t, StagingSlot>& stagingFree() {
   static auto* slots = new std::multimap<size_t, StagingSlot>;
   return *slots;
}
}  // namespace

HostFetch::HostFetch(const void* device_source, size_t bytes, void* stream) {
   capacity_ = 64u << 10;
   while (capacity_ < bytes) {
      capacity_ <<= 1;
   }
   {
      const std::lock_guard<std::mutex> lock(stagingMutex());
      const auto found = stagingFree().find(capacity_);
      if (found != stagingFree().end()) {
         host_ = found->second.host;
         event_ = found->second.event;
         stagingFree().erase(found);
      }
   }
   if (host_ == nullptr) {
      checkGpu(silo_gpu_host_alloc(capacity_, &host_), "silo_gpu_host_alloc");
      checkGpu(silo_gpu_event_create(&event_), "silo_gpu_event_create");
   }
   checkGpu(silo_gpu_memcpy_d2h_async(host_, device_source, bytes, stream), "silo_gpu_memcpy_d2h_async");
   checkGpu(silo_gpu_event_record(event_, stream), "silo_gpu_event_record");
}

HostFetch& HostFetch::operator=(HostFetch&& other) noexcept {
   if (this != &other) {
      this->~HostFetch();
      host_ = other.host_;
      event_ = other.event_;
      capacity_ = other.capacity_;
      other.host_ = nullptr;
      other.event_ = nullptr;
      other.capacity_ = 0;
   }
   return *this;
}

HostFetch::~HostFetch() {
   if (host_ != nullptr) {
      // the copy may still be in flight (an exception unwound the query): wait before the buffer is reused
      (void)silo_gpu_event_synchronize(event_);
      const std::lock_guard<std::mutex> lock(stagingMutex());
      stagingFree().emplace(capacity_, StagingSlot{host_, event_});
      host_ = nullptr;
   }
}

const void* HostFetch::wait() const {
   checkGpu(silo_gpu_event_synchronize(event_), "silo_gpu_event_synchronize");
   return host_;
}

// ---- RowSlot ---------------------------------------------------------------------------------------
namespace {
std::mutex& rowSlotMutex() {
   static std::mutex mutex;
   return mutex;
}
std::multimap<uint32_t, silo_gpu_row_slot*>& rowSlotFree() {
   static auto* free_list = new std::multimap<uint32_t, silo_gpu_row_slot*>();  // never destroyed: threads may outlive static destruction
   return *free_list;
}
}  // namespace

RowSlot::RowSlot(uint32_t row_capacity) : capacity_(row_capacity) {
   {
      const std::lock_guard<std::mutex> lock(rowSlotMutex());
      const auto found = rowSlotFree().find(row_capacity);
      if (found != rowSlotFree().end()) {
         slot_ = found->second;
         rowSlotFree().erase(found);
      }
   }
   if (slot_ == nullptr) {
      checkGpu(silo_gpu_row_slot_create(row_capacity, &slot_), "silo_gpu_row_slot_create");
   }
}

RowSlot& RowSlot::operator=(RowSlot&& other) noexcept {
   if (this != &other) {
      this->~RowSlot();
      slot_ = other.slot_;
      capacity_ = other.capacity_;
      stream_ = other.stream_;
      in_flight_ = other.in_flight_;
      other.slot_ = nullptr;
      other.in_flight_ = false;
   }
   return *this;
}

RowSlot::~RowSlot() {
   if (slot_ != nullptr) {
      if (in_flight_) {  // (an exception unwound the query) the kernel still writes into the slot: let it finish
         (void)silo_gpu_stream_synchronize(stream_);
      }
      const std::lock_guard<std::mutex> lock(rowSlotMutex());
      rowSlotFree().emplace(capacity_, slot_);
      slot_ = nullptr;
   }
}

void RowSlot::select(const uint32_t* counts, const uint8_t* reference_index, uint32_t n_positions, uint32_t n_symbols, double min_proportion, void* stream) {
   stream_ = stream;
   checkGpu(silo_gpu_mutations_select_to_slot(counts, reference_index, n_positions, n_symbols, min_proportion, slot_, stream), "silo_gpu_mutations_select_to_slot");
   in_flight_ = true;
}

std::pair<const silo_gpu_mutation_row*, uint32_t> RowSlot::wait() {
   const silo_gpu_mutation_row* rows = nullptr;
   uint32_t selected = 0;
   checkGpu(silo_gpu_row_slot_wait(slot_, &rows, &selected, stream_), "silo_gpu_row_slot_wait");
   in_flight_ = false;
   return {rows, selected};
}

// ---- pango lineage aliases (pango_lineage_alias.cpp:21-41, 88-102) -------------------------------
PangoLineageAliasLookup PangoLineageAliasLookup::fromJson(const json::Value& json) {
   std::unordered_map<std::string, std::vector<std::string>> alias_keys;
   for (const auto& [key, value] : json.members()) {
      if (value.is_array()) {
         std::vector<std::string> values;
         for (const auto& item : value.items()) {
            values.push_back(item.as_string());
         }
         alias_keys[key] = std::move(values);
      } else if (value.is_string() && !value.as_string().empty()) {
         alias_keys[key] = {value.as_string()};
      }
   }
   return PangoLineageAliasLookup(std::move(alias_keys));
}

std::string PangoLineageAliasLookup::unaliasPangoLineage(const std::string& pango_lineage) const {
   const auto dot = pango_lineage.find('.');
   const std::string prefix = pango_lineage.substr(0, dot);
   const auto found = alias_key.find(prefix);
   if (found == alias_key.end() || found->second.size() != 1) {
      return pango_lineage;
   }
   if (dot == std::string::npos) {
      return found->second.at(0);
   }
   // the reference reads the suffix through istream_iterator<char>, which skips whitespace
   std::string suffix;
   for (const char c : pango_lineage.substr(dot + 1)) {
      if (!std::isspace(static_cast<unsigned char>(c))) {
         suffix.push_back(c);
      }
   }
   return found->second.at(0) + '.' + suffix;
}

std::string PangoLineageAliasLookup::aliasPangoLineage(const std::string& pango_lineage) const {  // pango_lineage_alias.cpp:43-73
   std::vector<std::string> elements;
   std::string::size_type begin = 0;
   while (true) {
      const auto dot = pango_lineage.find('.', begin);
      elements.push_back(pango_lineage.substr(begin, dot == std::string::npos ? std::string::npos : dot - begin));
      if (dot == std::string::npos) {
         break;
      }
      begin = dot + 1;
   }
   const auto join = [&](size_t from, size_t to) {
      std::string out;
      for (size_t i = from; i < to; ++i) {
         if (i != from) {
            out.push_back('.');
         }
         out += elements[i];
      }
      return out;
   };
   const size_t num_elements = elements.size();
   // the longest proper prefix of at least three elements that some alias stands for
   for (size_t i = num_elements; i > 3; i--) {
      const std::string search_value = join(0, i - 1);
      for (const auto& [alias, alias_values] : alias_key) {
         if (alias_values.size() != 1) {
            continue;
         }
         if (alias_values.at(0) == search_value) {
            const std::string leftover_value = join(i - 1, num_elements);
            std::string value = alias;
            if (!leftover_value.empty()) {
               value += "." + leftover_value;
            }
            return value;
         }
      }
   }
   return pango_lineage;
}

std::vector<std::string> getParentLineages(const std::string& value) {  // pango_lineage.cpp:25-35
   std::vector<std::string> parent_lineages;
   std::string::size_type pos = 0;
   while (pos != std::string::npos) {
      pos = value.find('.', pos + 1);
      parent_lineages.push_back(value.substr(0, pos));
   }
   return parent_lineages;
}

// ---- pango lineage column ---------------------------------------------------------------------------
namespace storage::column {

PangoLineageColumnPartition::PangoLineageColumnPartition(const PangoLineageAliasLookup& alias_key, const DatabasePartition& partition)
    : alias_key(alias_key), partition(partition) {}

PangoLineageColumnPartition::~PangoLineageColumnPartition() {
   silo_gpu_free(d_value_ids_);
   for (auto& [key, ptr] : cache_) {
      silo_gpu_free(ptr);
   }
}

void PangoLineageColumnPartition::insert(const std::string& value) {  // pango_lineage_column.cpp:21-38
   const std::string resolved = alias_key.unaliasPangoLineage(value);
   auto found = lookup_unaliased_.find(resolved);
   if (found == lookup_unaliased_.end()) {
      found = lookup_unaliased_.emplace(resolved, static_cast<uint32_t>(dictionary_.size())).first;
      dictionary_.push_back(resolved);
   }
   value_ids_.push_back(found->second);
   ++n_rows_;
}

void PangoLineageColumnPartition::setValues(std::vector<std::string> dictionary, const uint32_t* value_ids, size_t n_rows) {
   dictionary_ = std::move(dictionary);
   lookup_unaliased_.clear();
   for (uint32_t id = 0; id < dictionary_.size(); ++id) {
      lookup_unaliased_.emplace(dictionary_[id], id);
   }
   value_ids_.assign(value_ids, value_ids + n_rows);
   n_rows_ = n_rows;
}

void PangoLineageColumnPartition::finalize() {
   if (d_value_ids_ != nullptr) {
      silo_gpu_free(d_value_ids_);
      d_value_ids_ = nullptr;
   }
   checkGpu(silo_gpu_upload_u32(value_ids_.data(), value_ids_.size(), &d_value_ids_), "silo_gpu_upload_u32");
   value_ids_.clear();
   value_ids_.shrink_to_fit();
}

std::optional<const uint64_t*> PangoLineageColumnPartition::lookup(const std::string& value, bool sublineages) const {
   const std::string resolved = alias_key.unaliasPangoLineage(value);
   const std::lock_guard<std::mutex> lock(mutex_);
   const auto key = std::make_pair(resolved, sublineages);
   if (const auto cached = cache_.find(key); cached != cache_.end()) {
      return cached->second;
   }
   // membership per dictionary entry: the entry itself, or (sublineages) every entry that has
   // `resolved` in its chain of dotted parents (pango_lineage_column.cpp:44-55).
   std::vector<uint8_t> membership(dictionary_.size(), 0);
   bool any = false;
   for (uint32_t id = 0; id < dictionary_.size(); ++id) {
      const std::string& entry = dictionary_[id];
      bool member = entry == resolved;
      if (!member && sublineages && entry.size() > resolved.size() && entry.compare(0, resolved.size(), resolved) == 0) {
         member = entry[resolved.size()] == '.' && !resolved.empty();
      }
      membership[id] = member ? 1 : 0;
      any = any || member;
   }
   if (!any || d_value_ids_ == nullptr) {
      return std::nullopt;  // unknown lineage -> Empty (pango_lineage_filter.cpp:55-57)
   }
   uint64_t* bitset = nullptr;
   checkGpu(silo_gpu_bitset_alloc(partition.store, &bitset), "silo_gpu_bitset_alloc");
   const int status = silo_gpu_bitset_from_value_ids(
      partition.store, bitset, d_value_ids_, membership.data(), static_cast<uint32_t>(membership.size()), nullptr
   );
   if (status != SILO_GPU_OK) {
      silo_gpu_free(bitset);
      checkGpu(status, "silo_gpu_bitset_from_value_ids");
   }
   cache_.emplace(key, bitset);
   return bitset;
}

std::optional<const uint64_t*> PangoLineageColumnPartition::filter(const std::string& value) const {
   return lookup(value, false);
}

std::optional<const uint64_t*> PangoLineageColumnPartition::filterIncludingSublineages(const std::string& value) const {
   return lookup(value, true);
}

}  // namespace storage::column

// ---- partitions / database ----------------------------------------------------------------------------
DatabasePartition::~DatabasePartition() {
   columns.pango_lineage_columns.clear();
   columns.metadata_columns.clear();
   columns.nuc_insertion_columns.clear();
   columns.aa_insertion_columns.clear();
   silo_gpu_store_destroy(store);
}

uint64_t& Database::queryFingerprint() {
   thread_local uint64_t fingerprint = 0;
   return fingerprint;
}

uint64_t Database::fingerprintOf(const std::string& query_text) {
   uint64_t hash = 0xCBF29CE484222325ull;
   for (const unsigned char c : query_text) {
      hash = (hash ^ c) * 0x100000001B3ull;
   }
   return hash;
}

Database::Timings& Database::lastTimings() {
   thread_local Timings timings;
   return timings;
}

void Database::applyStoreOptions() {
   for (DatabasePartition& partition : partitions) {
      checkGpu(silo_gpu_store_set_options(partition.store, &store_options), "silo_gpu_store_set_options");
   }
}

query_engine::QueryResult Database::executeQuery(const std::string& query) const {
   const query_engine::QueryEngine query_engine(*this);
   return query_engine.executeQuery(query);
}

std::string Database::executeQueryJson(const std::string& query) const {
   const query_engine::QueryEngine query_engine(*this);
   return query_engine.executeQueryJson(query);
}

std::pair<uint32_t, uint32_t> Database::positionWindow(size_t length) const {
   if (!shard_by_position || shard_world <= 1) {
      return {0, static_cast<uint32_t>(length)};
   }
   return {
      static_cast<uint32_t>(static_cast<uint64_t>(length) * shard_rank / shard_world),
      static_cast<uint32_t>(static_cast<uint64_t>(length) * (shard_rank + 1) / shard_world)};
}

uint32_t Database::ownerOfPosition(size_t position, size_t length) const {
   for (uint32_t rank = 0; rank < shard_world; ++rank) {
      const auto end = static_cast<uint64_t>(length) * (rank + 1) / shard_world;
      if (position < end) {
         return rank;
      }
   }
   return shard_world - 1;
}

void Database::setReferenceGenomes(const json::Value& reference_genomes) {  // reference_genomes.cpp
   nuc_sequences.clear();
   aa_sequences.clear();
   const auto read = [](const json::Value& list, auto& target, auto char_to_symbol, const char* what) {
      for (const auto& entry : list.items()) {
         const std::string& name = entry.at("name").as_string();
         const std::string& sequence = entry.at("sequence").as_string();
         auto& store = target[name];
         store.reference_sequence.reserve(sequence.size());
         for (const char c : sequence) {
            const auto symbol = char_to_symbol(c);
            if (!symbol.has_value()) {
               throw std::runtime_error(std::string("illegal character in the ") + what + " reference sequence " + name);
            }
            store.reference_sequence.push_back(*symbol);
         }
      }
   };
   read(reference_genomes.at("nucleotideSequences"), nuc_sequences, Nucleotide::charToSymbol, "nucleotide");
   read(reference_genomes.at("genes"), aa_sequences, AminoAcid::charToSymbol, "amino acid");
}

DatabasePartition& Database::addPartition(uint32_t sequence_count) {
   // one silo_gpu sequence store per nucleotide segment and per gene, in std::map (name) order
   std::vector<silo_gpu_seqstore_desc> descs;
   std::vector<std::vector<uint8_t>> references;
   static const std::vector<uint8_t> nuc_scan = [] {
      std::vector<uint8_t> out;
      for (const auto symbol : Nucleotide::VALID_MUTATION_SYMBOLS) {
         out.push_back(static_cast<uint8_t>(symbol));
      }
      return out;
   }();
   static const std::vector<uint8_t> aa_scan = [] {
      std::vector<uint8_t> out;
      for (const auto symbol : AminoAcid::VALID_MUTATION_SYMBOLS) {
         out.push_back(static_cast<uint8_t>(symbol));
      }
      return out;
   }();
   static const uint8_t nuc_extra[] = {static_cast<uint8_t>(Nucleotide::SYMBOL_MISSING)};
   static const uint8_t aa_extra[] = {static_cast<uint8_t>(AminoAcid::SYMBOL_MISSING)};

   const auto add = [&](const auto& stores, uint32_t alphabet, const std::vector<uint8_t>& scan, const uint8_t* extra) {
      for (const auto& [name, store] : stores) {
         auto& reference = references.emplace_back();
         const auto [window_begin, window_end] = positionWindow(store.reference_sequence.size());
         if (window_begin == window_end) {
            throw std::runtime_error("position-range shard of sequence store '" + name + "' is empty: more ranks than positions");
         }
         for (uint32_t position = window_begin; position < window_end; ++position) {
            reference.push_back(static_cast<uint8_t>(store.reference_sequence[position]));
         }
         silo_gpu_seqstore_desc desc{};
         desc.alphabet = alphabet;
         desc.positions = static_cast<uint32_t>(reference.size());
         desc.n_scan_symbols = static_cast<uint32_t>(scan.size());
         desc.scan_symbols = scan.data();
         desc.n_extra_symbols = 1;
         desc.extra_symbols = extra;
         descs.push_back(desc);
      }
   };
   add(nuc_sequences, SILO_GPU_ALPHABET_NUCLEOTIDE, nuc_scan, nuc_extra);
   add(aa_sequences, SILO_GPU_ALPHABET_AMINO_ACID, aa_scan, aa_extra);
   for (size_t i = 0; i < descs.size(); ++i) {
      descs[i].reference = references[i].data();
   }
   if (descs.empty()) {
      throw std::runtime_error("Database has no sequence stores: call setReferenceGenomes first");
   }
   silo_gpu_store_desc desc{};
   desc.device = device;
   desc.sequence_count = sequence_count;
   desc.n_seqstores = static_cast<uint32_t>(descs.size());
   desc.seqstores = descs.data();

   DatabasePartition& partition = partitions.emplace_back();
   partition.sequence_count = sequence_count;
   const int status = silo_gpu_store_create(&desc, &partition.store);
   if (status != SILO_GPU_OK) {
      partitions.pop_back();
      checkGpu(status, "silo_gpu_store_create");
   }
   checkGpu(silo_gpu_store_set_options(partition.store, &store_options), "silo_gpu_store_set_options");
   uint32_t seqstore_id = 0;
   for (const auto& [name, store] : nuc_sequences) {
      const auto [window_begin, window_end] = positionWindow(store.reference_sequence.size());
      partition.nuc_sequences.emplace(
         std::piecewise_construct, std::forward_as_tuple(name),
         std::forward_as_tuple(store.reference_sequence, partition.store, seqstore_id++, sequence_count, window_begin, window_end)
      );
   }
   for (const auto& [name, store] : aa_sequences) {
      const auto [window_begin, window_end] = positionWindow(store.reference_sequence.size());
      partition.aa_sequences.emplace(
         std::piecewise_construct, std::forward_as_tuple(name),
         std::forward_as_tuple(store.reference_sequence, partition.store, seqstore_id++, sequence_count, window_begin, window_end)
      );
   }
   return partition;
}

namespace {
template <typename SymbolType>
MutationTableLayout makeMutationTableLayout(const std::map<std::string, SequenceStore<SymbolType>>& stores, bool upload) {
   MutationTableLayout layout;
   std::vector<uint8_t> index;
   for (const auto& [name, store] : stores) {
      layout.position_offset[name] = layout.total_positions;
      layout.total_positions += static_cast<uint32_t>(store.reference_sequence.size());
      for (const auto reference_symbol : store.reference_sequence) {
         uint8_t found = 0xFF;
         for (size_t s = 0; s < SymbolType::VALID_MUTATION_SYMBOLS.size(); ++s) {
            if (SymbolType::VALID_MUTATION_SYMBOLS[s] == reference_symbol) {
               found = static_cast<uint8_t>(s);
            }
         }
         index.push_back(found);
      }
   }
   if (upload && !index.empty()) {
      void* device = nullptr;
      checkGpu(silo_gpu_upload_bytes(index.data(), index.size(), &device), "silo_gpu_upload_bytes");
      layout.reference_index_device = std::shared_ptr<const uint8_t>(static_cast<const uint8_t*>(device), [](const uint8_t* ptr) {
         silo_gpu_free(const_cast<uint8_t*>(ptr));
      });
   }
   return layout;
}
}  // namespace

void Database::appendMetadata(
   DatabasePartition& partition, const std::string& name, config::ColumnType type, const std::vector<std::string>& values
) {
   const auto known = database_config.getMetadata(name);
   if (!known.has_value()) {
      database_config.metadata.push_back({name, type});
   } else if (known->type != type) {
      throw std::runtime_error("metadata column '" + name + "' was declared with another type");
   }
   auto found = partition.columns.metadata_columns.find(name);
   if (found == partition.columns.metadata_columns.end()) {
      const bool is_sorted = type == config::ColumnType::DATE && database_config.date_to_sort_by.has_value() &&
                             *database_config.date_to_sort_by == name;
      found = partition.columns.metadata_columns
                 .emplace(std::piecewise_construct, std::forward_as_tuple(name), std::forward_as_tuple(type, is_sorted, &alias_key))
                 .first;
   }
   storage::column::MetadataColumnPartition& column = found->second;
   if (column.numRows() + values.size() > partition.sequence_count) {
      throw std::runtime_error("metadata column '" + name + "' holds more values than the partition has rows");
   }
   column.reserve(partition.sequence_count);
   if (type == config::ColumnType::NUC_INSERTION || type == config::ColumnType::AA_INSERTION) {
      // the index sees every entry; the column holds the standardised text (insertion_column.cpp:76-113).  Entries
      // without a sequence name belong to the default nucleotide sequence; amino-acid columns have no default
      // (database.cpp:73-80).
      const bool is_nucleotide = type == config::ColumnType::NUC_INSERTION;
      auto& insertion_columns = is_nucleotide ? partition.columns.nuc_insertion_columns : partition.columns.aa_insertion_columns;
      auto insertion_column = insertion_columns.find(name);
      if (insertion_column == insertion_columns.end()) {
         insertion_column = insertion_columns
                               .emplace(
                                  std::piecewise_construct, std::forward_as_tuple(name),
                                  std::forward_as_tuple(
                                     is_nucleotide ? std::optional<std::string>(database_config.default_nucleotide_sequence) : std::nullopt
                                  )
                               )
                               .first;
      }
      for (const std::string& value : values) {
         column.insert(insertion_column->second.insert(value, static_cast<uint32_t>(column.numRows())));
      }
      return;
   }
   for (const std::string& value : values) {
      column.insert(value);
   }
   if (type == config::ColumnType::INDEXED_PANGOLINEAGE) {
      auto lineage = partition.columns.pango_lineage_columns.find(name);
      if (lineage == partition.columns.pango_lineage_columns.end()) {
         lineage = partition.columns.pango_lineage_columns
                      .emplace(std::piecewise_construct, std::forward_as_tuple(name), std::forward_as_tuple(alias_key, partition))
                      .first;
      }
      for (const std::string& value : values) {
         lineage->second.insert(value);
      }
   }
}

void Database::appendUnalignedSequences(
   DatabasePartition& partition, const std::string& sequence_name, std::vector<std::optional<std::string>> values
) {
   if (nuc_sequences.count(sequence_name) == 0) {
      throw std::runtime_error("no nucleotide sequence named '" + sequence_name + "'");
   }
   auto& target = partition.unaligned_nuc_sequences[sequence_name];
   if (target.size() + values.size() > partition.sequence_count) {
      throw std::runtime_error("more unaligned sequences of '" + sequence_name + "' than the partition has rows");
   }
   for (auto& value : values) {
      target.push_back(std::move(value));
   }
}

void Database::finalize() {
   data_version = std::to_string(std::chrono::system_clock::to_time_t(std::chrono::system_clock::now()));  // DataVersion::mineDataVersion
   const bool device_in_use = !partitions.empty();
   nuc_mutation_layout = makeMutationTableLayout(nuc_sequences, device_in_use);
   aa_mutation_layout = makeMutationTableLayout(aa_sequences, device_in_use);
   for (auto& partition : partitions) {
      checkGpu(silo_gpu_store_finalize(partition.store), "silo_gpu_store_finalize");
      for (auto& [name, column] : partition.columns.pango_lineage_columns) {
         column.finalize();
      }
      for (auto& [name, column] : partition.columns.nuc_insertion_columns) {
         column.finalize();
      }
      for (auto& [name, column] : partition.columns.aa_insertion_columns) {
         column.finalize();
      }
      for (auto& [name, column] : partition.columns.metadata_columns) {
         if (column.numRows() != partition.sequence_count) {
            throw std::runtime_error("metadata column '" + name + "' does not have one value per row");
         }
         column.finalize();
      }
   }
}

}  // namespace silo
