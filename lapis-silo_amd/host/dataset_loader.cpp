#include "dataset_loader.h"

#include <dlfcn.h>

#include <algorithm>
#include <cstdio>
#include <cstring>
#include <filesystem>
#include <fstream>
#include <memory>
#include <optional>
#include <unordered_map>

#include "query_engine.h"

namespace silo::preprocessing {

namespace {

namespace fs = std::filesystem;

// ---- byte sources: plain / zstd / xz, the codecs resolved at run time (no link-time dependency) ---------
class ByteSource {
  public:
   virtual ~ByteSource() = default;
   virtual size_t read(char* destination, size_t capacity) = 0;  // 0 = end of stream
};

class PlainSource : public ByteSource {
  public:
   explicit PlainSource(const std::string& path) : file(std::fopen(path.c_str(), "rb")) {
      if (file == nullptr) {
         throw PreprocessingException("cannot open " + path);
      }
   }
   ~PlainSource() override { std::fclose(file); }
   size_t read(char* destination, size_t capacity) override { return std::fread(destination, 1, capacity, file); }

  private:
   std::FILE* file;
};

void* openLibrary(const char* const* names, const char* what) {
   for (const char* const* name = names; *name != nullptr; ++name) {
      if (void* handle = dlopen(*name, RTLD_NOW | RTLD_LOCAL); handle != nullptr) {
         return handle;
      }
   }
   throw PreprocessingException(std::string("cannot load the ") + what + " library needed to read a compressed input file");
}

template <typename Function>
Function resolve(void* library, const char* symbol) {
   void* address = dlsym(library, symbol);
   if (address == nullptr) {
      throw PreprocessingException(std::string("missing symbol ") + symbol);
   }
   return reinterpret_cast<Function>(address);
}

class ZstdSource : public ByteSource {  // zstd streaming API (stable since 1.0): zstd.h ZSTD_decompressStream
   struct InBuffer {
      const void* src;
      size_t size;
      size_t pos;
   };
   struct OutBuffer {
      void* dst;
      size_t size;
      size_t pos;
   };
   using CreateFn = void* (*)();
   using FreeFn = size_t (*)(void*);
   using InitFn = size_t (*)(void*);
   using DecompressFn = size_t (*)(void*, OutBuffer*, InBuffer*);
   using IsErrorFn = unsigned (*)(size_t);

  public:
   explicit ZstdSource(const std::string& path) : input(path), compressed(1 << 17) {
      static const char* const names[] = {"libzstd.so.1", "libzstd.so", "/opt/conda/lib/libzstd.so", nullptr};
      void* library = openLibrary(names, "zstd");
      release = resolve<FreeFn>(library, "ZSTD_freeDStream");
      decompress = resolve<DecompressFn>(library, "ZSTD_decompressStream");
      is_error = resolve<IsErrorFn>(library, "ZSTD_isError");
      stream = resolve<CreateFn>(library, "ZSTD_createDStream")();
      if (stream == nullptr || is_error(resolve<InitFn>(library, "ZSTD_initDStream")(stream)) != 0) {
         throw PreprocessingException("cannot initialise the zstd decoder");
      }
   }
   ~ZstdSource() override { release(stream); }

   size_t read(char* destination, size_t capacity) override {
      OutBuffer out{destination, capacity, 0};
      while (out.pos == 0) {
         if (in.pos == in.size) {
            const size_t got = input.read(compressed.data(), compressed.size());
            if (got == 0) {
               return 0;
            }
            in = {compressed.data(), got, 0};
         }
         if (is_error(decompress(stream, &out, &in)) != 0) {
            throw PreprocessingException("corrupt zstd stream");
         }
      }
      return out.pos;
   }

  private:
   PlainSource input;
   std::vector<char> compressed;
   InBuffer in{nullptr, 0, 0};
   void* stream = nullptr;
   FreeFn release = nullptr;
   DecompressFn decompress = nullptr;
   IsErrorFn is_error = nullptr;
};

class XzSource : public ByteSource {  // liblzma 5.x: lzma_stream_decoder / lzma_code (lzma/base.h layout)
   struct Stream {
      const uint8_t* next_in;
      size_t avail_in;
      uint64_t total_in;
      uint8_t* next_out;
      size_t avail_out;
      uint64_t total_out;
      const void* allocator;
      void* internal;
      void* reserved_ptr1;
      void* reserved_ptr2;
      void* reserved_ptr3;
      void* reserved_ptr4;
      uint64_t reserved_int1;
      uint64_t reserved_int2;
      size_t reserved_int3;
      size_t reserved_int4;
      int reserved_enum1;
      int reserved_enum2;
   };
   using DecoderFn = int (*)(Stream*, uint64_t, uint32_t);
   using CodeFn = int (*)(Stream*, int);
   using EndFn = void (*)(Stream*);

  public:
   explicit XzSource(const std::string& path) : input(path), compressed(1 << 17) {
      static const char* const names[] = {"liblzma.so.5", "liblzma.so", "/opt/conda/lib/liblzma.so", nullptr};
      void* library = openLibrary(names, "lzma");
      code = resolve<CodeFn>(library, "lzma_code");
      end = resolve<EndFn>(library, "lzma_end");
      memset(&stream, 0, sizeof(stream));
      if (resolve<DecoderFn>(library, "lzma_stream_decoder")(&stream, UINT64_MAX, /*LZMA_CONCATENATED*/ 0x08) != 0) {
         throw PreprocessingException("cannot initialise the xz decoder");
      }
   }
   ~XzSource() override { end(&stream); }

   size_t read(char* destination, size_t capacity) override {
      if (finished) {
         return 0;
      }
      stream.next_out = reinterpret_cast<uint8_t*>(destination);
      stream.avail_out = capacity;
      while (stream.avail_out == capacity) {
         if (stream.avail_in == 0 && !input_done) {
            const size_t got = input.read(compressed.data(), compressed.size());
            input_done = got == 0;
            stream.next_in = reinterpret_cast<const uint8_t*>(compressed.data());
            stream.avail_in = got;
         }
         const int status = code(&stream, input_done ? /*LZMA_FINISH*/ 3 : /*LZMA_RUN*/ 0);
         if (status == /*LZMA_STREAM_END*/ 1) {
            finished = true;
            break;
         }
         if (status != /*LZMA_OK*/ 0) {
            throw PreprocessingException("corrupt xz stream");
         }
      }
      return capacity - stream.avail_out;
   }

  private:
   PlainSource input;
   std::vector<char> compressed;
   Stream stream{};
   CodeFn code = nullptr;
   EndFn end = nullptr;
   bool input_done = false;
   bool finished = false;
};

bool endsWith(const std::string& text, const std::string& suffix) {
   return text.size() >= suffix.size() && text.compare(text.size() - suffix.size(), suffix.size(), suffix) == 0;
}

std::unique_ptr<ByteSource> openSource(const std::string& path) {
   if (endsWith(path, ".zst")) {
      return std::make_unique<ZstdSource>(path);
   }
   if (endsWith(path, ".xz")) {
      return std::make_unique<XzSource>(path);
   }
   return std::make_unique<PlainSource>(path);
}

class LineReader {
  public:
   explicit LineReader(const std::string& path) : source(openSource(path)), buffer(1 << 20) {}

   bool next(std::string& line) {
      line.clear();
      while (true) {
         if (begin == end) {
            end = source->read(buffer.data(), buffer.size());
            begin = 0;
            if (end == 0) {
               return !line.empty();
            }
         }
         const char* start = buffer.data() + begin;
         const void* newline = memchr(start, '\n', end - begin);
         if (newline != nullptr) {
            const size_t length = static_cast<const char*>(newline) - start;
            line.append(start, length);
            begin += length + 1;
            if (!line.empty() && line.back() == '\r') {
               line.pop_back();
            }
            return true;
         }
         line.append(start, end - begin);
         begin = end;
      }
   }

  private:
   std::unique_ptr<ByteSource> source;
   std::vector<char> buffer;
   size_t begin = 0;
   size_t end = 0;
};

std::string readWholeFile(const std::string& path) {
   std::ifstream stream(path, std::ios::binary);
   if (!stream) {
      throw PreprocessingException("cannot open " + path);
   }
   return {std::istreambuf_iterator<char>(stream), std::istreambuf_iterator<char>()};
}

/// `<name>`, `<name>.zst` or `<name>.xz`, whichever exists.
std::optional<std::string> findWithCompression(const fs::path& base) {
   for (const char* suffix : {"", ".zst", ".xz"}) {
      const fs::path candidate = base.string() + suffix;
      if (fs::exists(candidate)) {
         return candidate.string();
      }
   }
   return std::nullopt;
}

// ---- the two YAML files: a flat key/value file and the schema (only the keys the path needs) -------------
std::string unquote(std::string value) {
   const auto first = value.find_first_not_of(" \t");
   const auto last = value.find_last_not_of(" \t\r");
   if (first == std::string::npos) {
      return "";
   }
   value = value.substr(first, last - first + 1);
   if (value.size() >= 2 && ((value.front() == '"' && value.back() == '"') || (value.front() == '\'' && value.back() == '\''))) {
      value = value.substr(1, value.size() - 2);
   }
   return value;
}

std::unordered_map<std::string, std::string> readFlatYaml(const std::string& path) {
   std::unordered_map<std::string, std::string> out;
   std::ifstream stream(path);
   std::string line;
   while (std::getline(stream, line)) {
      const auto colon = line.find(':');
      if (colon == std::string::npos || line.find_first_not_of(" \t") == std::string::npos || line[line.find_first_not_of(" \t")] == '#') {
         continue;
      }
      const std::string value = unquote(line.substr(colon + 1));
      if (value.empty() && unquote(line.substr(colon + 1) + "x").size() <= 1) {
         continue;  // "key:" without a value is YAML null — not set; `key: ""` below is an (empty) string and is kept
      }
      out[unquote(line.substr(0, colon))] = value;
   }
   return out;
}

struct DatabaseSchema {
   std::string instance_name;
   std::string primary_key;
   std::optional<std::string> date_to_sort_by;
   std::optional<std::string> partition_by;
   std::optional<std::string> default_nucleotide_sequence;
   struct Column {
      std::string name;
      std::string type;
      bool generate_index = false;
      bool generate_index_given = false;
   };
   std::vector<Column> metadata;  // in file order
};

/// config_repository.cpp:22-108 (validateConfig): the rules a database config has to satisfy, with the reference's messages.
void validateDatabaseSchema(const DatabaseSchema& schema) {
   std::map<std::string, std::string> type_of;
   for (const auto& column : schema.metadata) {
      if (type_of.count(column.name) != 0) {
         throw PreprocessingException("Metadata " + column.name + " is defined twice in the config");
      }
      if (column.generate_index && column.type != "string" && column.type != "pango_lineage") {
         throw PreprocessingException(
            "Metadata '" + column.name + "' generate_index is set, but generating an index is only allowed for types STRING and PANGOLINEAGE"
         );
      }
      if (!column.generate_index && column.type == "pango_lineage") {
         throw PreprocessingException(
            "Metadata '" + column.name + "' generate_index is not set, but generating an index is mandatory for type PANGOLINEAGE"
         );
      }
      type_of[column.name] = column.type;
   }
   if (schema.metadata.empty()) {
      throw PreprocessingException("Database config without fields not possible");
   }
   if (type_of.count(schema.primary_key) == 0) {
      throw PreprocessingException("Primary key is not in metadata");
   }
   if (schema.date_to_sort_by.has_value()) {
      const auto found = type_of.find(*schema.date_to_sort_by);
      if (found == type_of.end()) {
         throw PreprocessingException("date_to_sort_by '" + *schema.date_to_sort_by + "' is not in metadata");
      }
      if (found->second != "date") {
         throw PreprocessingException("date_to_sort_by '" + *schema.date_to_sort_by + "' must be of type DATE");
      }
   }
   if (schema.partition_by.has_value()) {
      const auto found = type_of.find(*schema.partition_by);
      if (found == type_of.end()) {
         throw PreprocessingException("partition_by '" + *schema.partition_by + "' is not in metadata");
      }
      if (found->second != "pango_lineage") {
         throw PreprocessingException("partition_by '" + *schema.partition_by + "' must be of type PANGOLINEAGE");
      }
   }
}

DatabaseSchema readDatabaseConfig(const std::string& path) {  // database_config.cpp:47-90, 197-231
   DatabaseSchema schema;
   std::ifstream stream(path);
   if (!stream) {
      throw PreprocessingException("Failed to read database config: cannot open " + path);
   }
   std::string line;
   std::string open_list;      // the key whose (block) value the following "- " items belong to
   size_t list_indent = 0;
   bool has_schema = false, has_metadata = false, has_primary_key = false;
   while (std::getline(stream, line)) {
      const auto first = line.find_first_not_of(" \t");
      if (first == std::string::npos || line[first] == '#') {
         continue;
      }
      const auto colon = line.find(':');
      if (colon == std::string::npos) {
         continue;
      }
      std::string key = unquote(line.substr(0, colon));
      const std::string value = unquote(line.substr(colon + 1));
      bool starts_item = false;
      if (!key.empty() && key.front() == '-') {
         key = unquote(key.substr(1));
         starts_item = true;
      } else if (value.empty()) {
         if (key == "schema") {
            has_schema = true;
         } else {
            open_list = key;  // "metadata:", "features:"
            list_indent = first;
            has_metadata = has_metadata || key == "metadata";
         }
         continue;
      } else if (!open_list.empty() && first <= list_indent) {
         open_list.clear();  // a scalar at the level of the list's key ends the list
      }
      const bool in_metadata = open_list == "metadata";
      if (starts_item && in_metadata) {
         schema.metadata.emplace_back();
      }
      if (open_list.empty() || !in_metadata) {
         if (key == "instanceName" && open_list.empty()) {
            schema.instance_name = value;
         } else if (key == "primaryKey" && open_list.empty()) {
            schema.primary_key = value;
            has_primary_key = true;
         } else if (key == "dateToSortBy" && open_list.empty()) {
            schema.date_to_sort_by = value;
         } else if (key == "partitionBy" && open_list.empty()) {
            schema.partition_by = value;
         } else if (key == "defaultNucleotideSequence") {
            schema.default_nucleotide_sequence = value;
         }
         continue;  // entries of other lists ("features") are none of ours (database_config.test.cpp:143-147)
      }
      if (key == "name") {
         schema.metadata.back().name = value;
      } else if (key == "type") {
         schema.metadata.back().type = value;
      } else if (key == "generateIndex") {
         schema.metadata.back().generate_index = value == "true";
         schema.metadata.back().generate_index_given = true;
      }
   }
   for (auto& column : schema.metadata) {
      if (!column.generate_index_given) {
         column.generate_index = column.type == "pango_lineage";  // the reference's default (database_config.cpp:138-142)
      }
   }
   if (!has_schema || !has_metadata || !has_primary_key) {
      // what yaml-cpp reports when the reference decodes a schema without the key (database_config.test.cpp:149-160)
      const std::string missing = !has_schema ? "schema" : (!has_metadata ? "metadata" : "primaryKey");
      throw PreprocessingException("database config " + path + ": invalid node; first invalid key: \"" + missing + "\"");
   }
   for (const auto& column : schema.metadata) {
      if (column.name.empty() || !config::columnTypeFromConfig(column.type, column.generate_index).has_value()) {
         throw PreprocessingException("database config " + path + ": metadata entry '" + column.name + "' has no valid name / type ('" + column.type + "')");
      }
   }
   return schema;
}

/// Rows of the metadata columns, staged as text and handed to the database in batches.
class MetadataWriter {
  public:
   MetadataWriter(Database& database, DatabasePartition& partition, const DatabaseSchema& schema) : database(database), partition(partition) {
      for (const auto& column : schema.metadata) {
         columns.push_back({column.name, *config::columnTypeFromConfig(column.type, column.generate_index), {}});
      }
   }
   [[nodiscard]] size_t size() const { return columns.size(); }
   [[nodiscard]] const std::string& name(size_t column) const { return columns[column].name; }
   void add(size_t column, std::string value) { columns[column].values.push_back(std::move(value)); }
   void rowDone() {
      if (!columns.empty() && columns.front().values.size() >= BATCH) {
         flush();
      }
   }
   void flush() {
      for (auto& column : columns) {
         database.appendMetadata(partition, column.name, column.type, column.values);
         column.values.clear();
      }
   }

  private:
   static constexpr size_t BATCH = 65536;
   struct Column {
      std::string name;
      config::ColumnType type;
      std::vector<std::string> values;
   };
   Database& database;
   DatabasePartition& partition;
   std::vector<Column> columns;
};

// ---- staging of one sequence store: batches of equal-length rows appended to the device ---------------------
class StoreWriter {
  public:
   StoreWriter(silo_gpu_store* store, uint32_t seqstore_id, size_t length, std::string name)
       : store(store), seqstore_id(seqstore_id), length(length), name(std::move(name)) {}

   void add(const std::string* sequence, DatasetSummary& summary) {
      if (sequence == nullptr) {
         is_null.push_back(1);
         chars.resize(chars.size() + length, 'N');
         ++summary.null_sequences;
      } else {
         if (sequence->size() != length) {
            throw PreprocessingException(
               "sequence " + std::to_string(next_row + is_null.size()) + " of '" + name + "' has length " +
               std::to_string(sequence->size()) + ", the reference has " + std::to_string(length)
            );
         }
         is_null.push_back(0);
         chars.insert(chars.end(), sequence->begin(), sequence->end());
      }
      if (is_null.size() >= BATCH) {
         flush();
      }
   }

   /// Two-pass build (silo_gpu_store_build_pass): 1 before the rows are fed the first time (they are only counted), 2 before
   /// they are fed again (they are written straight into the adaptive planes); the row cursor starts over.
   void beginPass(int pass) {
      flush();
      checkGpu(silo_gpu_store_build_pass(store, seqstore_id, pass), "silo_gpu_store_build_pass");
      next_row = 0;
   }

   void flush() {
      if (is_null.empty()) {
         return;
      }
      checkGpu(
         silo_gpu_store_append_sequences(
            store, seqstore_id, next_row, static_cast<uint32_t>(is_null.size()), chars.data(), is_null.data()
         ),
         "silo_gpu_store_append_sequences"
      );
      next_row += static_cast<uint32_t>(is_null.size());
      is_null.clear();
      chars.clear();
   }

  private:
   static constexpr size_t BATCH = 4096;  // the reference buffers 1024 genomes (sequence_store.cpp:34)
   silo_gpu_store* store;
   uint32_t seqstore_id;
   size_t length;
   std::string name;
   uint32_t next_row = 0;
   std::vector<char> chars;
   std::vector<uint8_t> is_null;
};

std::vector<std::string> splitTabs(const std::string& line) {
   std::vector<std::string> out;
   size_t begin = 0;
   while (true) {
      const auto tab = line.find('\t', begin);
      out.push_back(line.substr(begin, tab == std::string::npos ? std::string::npos : tab - begin));
      if (tab == std::string::npos) {
         return out;
      }
      begin = tab + 1;
   }
}

std::unordered_map<std::string, std::string> readFasta(const std::string& path) {  // common/fasta_reader.cpp:11-47
   // The reference reads strict two-line records (key line, genome line) and throws FastaFormatException for a key line
   // without '>' or a key without a genome line; sequences wrapped over several lines are accepted here in addition.
   std::unordered_map<std::string, std::string> out;
   LineReader reader(path);
   std::string line;
   std::string* current = nullptr;
   std::string current_key;
   bool current_has_genome = true;
   while (reader.next(line)) {
      if (line.empty()) {
         continue;
      }
      if (line.front() == '>') {
         if (!current_has_genome) {
            throw PreprocessingException("Missing genome sequence in line following key: " + current_key);
         }
         current_key = line.substr(1);
         current = &out[current_key];
         current->clear();
         current_has_genome = false;
      } else if (current != nullptr) {
         current->append(line);
         current_has_genome = true;
      } else {
         throw PreprocessingException("Fasta key prefix '>' missing for key: " + line);
      }
   }
   if (!current_has_genome) {
      throw PreprocessingException("Missing genome sequence in line following key: " + current_key);
   }
   return out;
}

}  // namespace

std::string describeFasta(const std::string& path) {
   std::vector<std::pair<std::string, std::string>> records;
   for (auto& [key, genome] : readFasta(path)) {
      records.emplace_back(key, std::move(genome));
   }
   std::sort(records.begin(), records.end());
   json::Value::Array out;
   for (const auto& [key, genome] : records) {
      out.emplace_back(json::Value::Array{json::Value(key), json::Value(genome)});
   }
   return json::Value(std::move(out)).dump();
}

std::string describeDatabaseConfig(const std::string& path, bool validate) {
   const DatabaseSchema schema = readDatabaseConfig(path);
   if (validate) {
      validateDatabaseSchema(schema);
   }
   json::Value::Object out;
   const auto optional_text = [](const std::optional<std::string>& value) { return value.has_value() ? json::Value(*value) : json::Value(nullptr); };
   out.insertOrAssign("instanceName", schema.instance_name);
   out.insertOrAssign("primaryKey", schema.primary_key);
   out.insertOrAssign("dateToSortBy", optional_text(schema.date_to_sort_by));
   out.insertOrAssign("partitionBy", optional_text(schema.partition_by));
   json::Value::Array columns;
   for (const auto& column : schema.metadata) {
      json::Value::Object entry;
      entry.insertOrAssign("name", column.name);
      entry.insertOrAssign("type", column.type);
      entry.insertOrAssign("generateIndex", column.generate_index);
      columns.emplace_back(std::move(entry));
   }
   out.insertOrAssign("metadata", std::move(columns));
   return json::Value(std::move(out)).dump();
}

/// The order the reference lays rows out in: it partitions by the partitionBy column — the keys in ascending order, consecutive
/// keys merged into <= 32 partitions (preprocessor.cpp:159-227) — and orders the rows of a partition by dateToSortBy, then by
/// the primary key (database_config.cpp:190-198).  One partition here, so the rows go by (partition key, date, primary key):
/// every lineage is a row range, and inside it the dates ascend — lineage and date filters then select runs of rows, whose
/// column tiles and key slices are the only ones a scan reads.  Rows without a date come last (DuckDB's NULLS LAST).
std::vector<uint32_t> referenceRowOrder(
   const std::vector<std::string>& partition_keys, const std::vector<std::string>& dates, const std::vector<std::string>& primary_keys
) {
   std::vector<uint32_t> order(primary_keys.size());
   for (size_t row = 0; row < order.size(); ++row) {
      order[row] = static_cast<uint32_t>(row);
   }
   std::stable_sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) {
      if (!partition_keys.empty() && partition_keys[a] != partition_keys[b]) {
         return partition_keys[a] < partition_keys[b];
      }
      if (!dates.empty() && dates[a] != dates[b]) {
         return dates[b].empty() || (!dates[a].empty() && dates[a] < dates[b]);
      }
      return primary_keys[a] < primary_keys[b];
   });
   return order;
}

DatasetSummary loadDataset(Database& database, const std::string& directory) {
   if (!database.partitions.empty()) {
      throw PreprocessingException("loadDataset needs an empty database");
   }
   const fs::path root(directory);
   std::unordered_map<std::string, std::string> config;
   if (fs::exists(root / "preprocessing_config.yaml")) {
      config = readFlatYaml((root / "preprocessing_config.yaml").string());
   }
   const auto setting = [&](const char* key, const char* fallback) {
      const auto found = config.find(key);
      // a key that is present wins even when it is the empty string: `nucleotideSequencePrefix: ""` names the files
      // `<sequence>.fasta` (preprocessing_config_reader.test.cpp:35-50)
      return found != config.end() ? found->second : std::string(fallback);
   };
   if (config.count("ndjsonInputFilename") != 0 && config.count("metadataFilename") != 0) {  // preprocessing_config_reader.cpp:54-60
      throw PreprocessingException(
         "Cannot specify both a ndjsonInputFilename ('" + config["ndjsonInputFilename"] + "') and metadataFilename('" +
         config["metadataFilename"] + "')."
      );
   }
   const DatabaseSchema schema = readDatabaseConfig((root / "database_config.yaml").string());
   validateDatabaseSchema(schema);  // ConfigRepository::getValidatedConfig
   if (schema.default_nucleotide_sequence.has_value()) {
      database.database_config.default_nucleotide_sequence = *schema.default_nucleotide_sequence;
   }
   database.setReferenceGenomes(json::parse(readWholeFile((root / setting("referenceGenomeFilename", "reference_genomes.json")).string())));
   const fs::path alias_path = root / setting("pangoLineageDefinitionFilename", "pangolineage_alias.json");
   if (fs::exists(alias_path)) {
      database.alias_key = PangoLineageAliasLookup::fromJson(json::parse(readWholeFile(alias_path.string())));
   }

   DatasetSummary summary;
   summary.nucleotide_stores = database.nuc_sequences.size();
   summary.amino_acid_stores = database.aa_sequences.size();
   for (const auto& column : schema.metadata) {
      summary.lineage_columns += column.type == "pango_lineage" ? 1 : 0;
   }
   database.database_config.primary_key = schema.primary_key;
   database.database_config.date_to_sort_by = schema.date_to_sort_by;

   const bool from_ndjson = config.count("ndjsonInputFilename") != 0;
   const std::string input_path = (root / (from_ndjson ? config["ndjsonInputFilename"] : setting("metadataFilename", "metadata.tsv"))).string();

   // "sortRows: false" in preprocessing_config.yaml keeps the rows in file order (no reference analogue; the default lays them out
   // as the reference does, see referenceRowOrder)
   const bool sort_rows = setting("sortRows", "true") != "false";
   // pass 1: the row count (a device store is allocated for a known number of rows)
   size_t rows = 0;
   {
      LineReader reader(input_path);
      std::string line;
      while (reader.next(line)) {
         rows += line.find_first_not_of(" \t") != std::string::npos ? 1 : 0;
      }
      if (!from_ndjson && rows > 0) {
         --rows;  // TSV header
      }
   }
   summary.sequence_count = rows;
   if (rows > UINT32_MAX) {
      throw PreprocessingException("more than 2^32 rows in one partition");
   }
   DatabasePartition& partition = database.addPartition(static_cast<uint32_t>(rows));
   MetadataWriter metadata_writer(database, partition, schema);
   std::vector<std::pair<std::string, StoreWriter>> nuc_writers;
   std::vector<std::pair<std::string, StoreWriter>> aa_writers;
   for (const auto& [name, store] : partition.nuc_sequences) {
      nuc_writers.emplace_back(name, StoreWriter(partition.store, store.seqstore_id, store.reference_sequence.size(), name));
   }
   for (const auto& [name, store] : partition.aa_sequences) {
      aa_writers.emplace_back(name, StoreWriter(partition.store, store.seqstore_id, store.reference_sequence.size(), name));
   }

   // "twoPassBuild: true" in preprocessing_config.yaml (no reference analogue): every sequence store is fed twice — counted,
   // then written straight into its adaptive planes — for inputs whose build-time planes would not fit beside the finished store
   const bool two_pass = setting("twoPassBuild", "false") == "true";
   if (from_ndjson) {  // preprocessor.cpp:87-131: one JSON object per line
      LineReader reader(input_path);
      std::string line;
      bool first_record = true;
      // One JSON object per record; normally one per line, but the reference's reader (DuckDB read_json) also takes objects
      // spread over several lines (testBaseData/ndjsonFiles/oneline_*.json.zst): lines are joined until the braces balance.
      const auto next_record = [](LineReader& reader, std::string& record_text) {
         record_text.clear();
         std::string part;
         int depth = 0;
         bool in_string = false, escaped = false, seen_value = false;
         while (reader.next(part)) {
            if (record_text.empty() && part.find_first_not_of(" \t\r") == std::string::npos) {
               continue;
            }
            if (record_text.empty() && part.size() > 2 && part.front() == '{' && part.back() == '}') {
               record_text.swap(part);  // the usual case, one object per line: no need to look inside 30 kb of sequence text
               return true;
            }
            for (const char c : part) {
               if (in_string) {
                  if (escaped) {
                     escaped = false;
                  } else if (c == '\\') {
                     escaped = true;
                  } else if (c == '"') {
                     in_string = false;
                  }
               } else if (c == '"') {
                  in_string = true;
               } else if (c == '{' || c == '[') {
                  ++depth;
                  seen_value = true;
               } else if (c == '}' || c == ']') {
                  --depth;
               }
            }
            record_text += part;
            record_text += '\n';
            if (depth <= 0 && (seen_value || !in_string)) {
               return true;
            }
         }
         return !record_text.empty();
      };
      // the aligned sequences of one record, to the writers of every sequence store
      const auto feed_sequences = [&](const json::Value& record, DatasetSummary& target) {
         static const json::Value missing_section;
         const auto feed = [&](std::vector<std::pair<std::string, StoreWriter>>& writers, const char* section) {
            const json::Value& sequences = record.contains(section) ? record[section] : missing_section;
            for (auto& [name, writer] : writers) {
               if (sequences.contains(name) && sequences[name].is_string()) {
                  writer.add(&sequences[name].as_string(), target);
               } else {
                  writer.add(nullptr, target);  // null genome: missing at every position (sequence_store.cpp:166-169)
               }
            }
         };
         feed(nuc_writers, "alignedNucleotideSequences");
         feed(aa_writers, "alignedAminoAcidSequences");
      };
      if (two_pass) {  // the file is read twice: the first time the sequences are only counted
         const auto begin_pass = [&](int pass) {
            for (auto& [name, writer] : nuc_writers) {
               writer.beginPass(pass);
            }
            for (auto& [name, writer] : aa_writers) {
               writer.beginPass(pass);
            }
         };
         begin_pass(1);
         LineReader counting_reader(input_path);
         DatasetSummary counted_before;
         while (next_record(counting_reader, line)) {
            feed_sequences(json::parse(line), counted_before);
         }
         begin_pass(2);
      }
      // The records in the reference's row order: the file is held in memory (its texts, while they fit `sortRowsMaxBytes`,
      // 8 GiB by default), the sort keys read from every record's metadata, and the records then fed in that order — the
      // reference leaves the same to DuckDB.  A larger file keeps its own order.
      std::vector<std::string> held;
      std::vector<uint32_t> order;
      if (sort_rows) {
         const uint64_t budget = std::strtoull(setting("sortRowsMaxBytes", "8589934592").c_str(), nullptr, 10);
         uint64_t bytes = 0;
         LineReader holding_reader(input_path);
         std::string text;
         bool fits = true;
         while (fits && next_record(holding_reader, text)) {
            bytes += text.size();
            fits = bytes <= budget;
            held.push_back(std::move(text));
            text.clear();
         }
         if (!fits) {
            held.clear();
         } else {
            std::vector<std::string> partition_keys, dates, primary_keys;
            for (const std::string& text_of_record : held) {
               const json::Value record = json::parse(text_of_record);
               static const json::Value no_metadata = json::Value::object();
               const json::Value& metadata = record.contains("metadata") && record["metadata"].is_object() ? record["metadata"] : no_metadata;
               const auto text_of = [&](const std::string& column) {
                  return metadata.contains(column) && metadata[column].is_string() ? metadata[column].as_string() : std::string();
               };
               if (schema.partition_by.has_value()) {
                  partition_keys.push_back(text_of(*schema.partition_by));
               }
               if (schema.date_to_sort_by.has_value()) {
                  dates.push_back(text_of(*schema.date_to_sort_by));
               }
               primary_keys.push_back(text_of(schema.primary_key));
            }
            order = referenceRowOrder(partition_keys, dates, primary_keys);
         }
      }
      size_t next_held = 0;
      const auto next_in_order = [&](std::string& record_text) {
         if (held.empty()) {
            return next_record(reader, record_text);
         }
         if (next_held == order.size()) {
            return false;
         }
         record_text = held[order[next_held++]];
         return true;
      };
      while (next_in_order(line)) {
         const json::Value record = json::parse(line);
         if (first_record) {
            // sequence_info.cpp:91-157 (SequenceInfo::validate): the sequence names of the FIRST record and of the
            // reference genomes have to be the same sets
            first_record = false;
            const auto validate_names = [&](const char* section, const char* kind, const std::vector<std::pair<std::string, StoreWriter>>& writers) {
               std::vector<std::string> in_file;
               if (record.contains(section) && record[section].is_object()) {
                  for (const auto& [name, value] : record[section].members()) {
                     in_file.push_back(name);
                  }
               }
               for (const std::string& name : in_file) {
                  if (std::none_of(writers.begin(), writers.end(), [&](const auto& writer) { return writer.first == name; })) {
                     throw PreprocessingException(
                        std::string("The aligned ") + kind + " sequence " + name + " which is contained in the input file " + input_path +
                        " is not contained in the reference sequences."
                     );
                  }
               }
               for (const auto& [name, writer] : writers) {
                  if (std::find(in_file.begin(), in_file.end(), name) == in_file.end()) {
                     throw PreprocessingException(
                        std::string("The aligned ") + kind + " sequence " + name + " which is contained in the reference sequences is not contained in the input file " +
                        input_path + "."
                     );
                  }
               }
            };
            validate_names("alignedNucleotideSequences", "nucleotide", nuc_writers);
            validate_names("alignedAminoAcidSequences", "amino acid", aa_writers);
            // metadata_info.cpp:13-47,121-163 (validateFromNdjsonFile): every configured column is a key of the first record's
            // metadata (or one of the two top-level insertion maps)
            for (size_t k = 0; k < metadata_writer.size(); ++k) {
               const std::string& column = metadata_writer.name(k);
               const bool in_metadata = record.contains("metadata") && record["metadata"].is_object() && record["metadata"].contains(column);
               const bool insertion_map = (column == "nucleotideInsertions" || column == "aminoAcidInsertions") && record.contains(column);
               if (!in_metadata && !insertion_map) {
                  throw PreprocessingException("The metadata field '" + column + "' which is contained in the database config is not contained in the input.");
               }
            }
         }
         const json::Value& metadata = record.at("metadata");
         for (size_t k = 0; k < metadata_writer.size(); ++k) {
            const std::string& column = metadata_writer.name(k);
            if ((column == "nucleotideInsertions" || column == "aminoAcidInsertions") && record.contains(column) && record[column].is_object()) {
               // metadata_info.cpp:61-94: the top-level maps {sequence: [entries]} become one column value
               // "sequence:entry,sequence:entry,..." (list_string_agg of the flattened, prefixed lists)
               std::string joined;
               for (const auto& [sequence_name, entries] : record[column].members()) {
                  for (const auto& entry : entries.items()) {
                     if (entry.is_string()) {
                        joined += (joined.empty() ? "" : ",") + sequence_name + ":" + entry.as_string();
                     }
                  }
               }
               metadata_writer.add(k, std::move(joined));
               continue;
            }
            if (!metadata.contains(column) || metadata[column].is_null()) {
               metadata_writer.add(k, "");
            } else if (metadata[column].is_string()) {
               metadata_writer.add(k, metadata[column].as_string());
            } else {
               metadata_writer.add(k, metadata[column].dump());  // numbers in their JSON text form
            }
         }
         metadata_writer.rowDone();
         feed_sequences(record, summary);
         if (record.contains("unalignedNucleotideSequences") && record["unalignedNucleotideSequences"].is_object()) {
            const json::Value& unaligned = record["unalignedNucleotideSequences"];
            for (const auto& [name, store] : partition.nuc_sequences) {
               std::vector<std::optional<std::string>> value(1);
               if (unaligned.contains(name) && unaligned[name].is_string()) {
                  value[0] = unaligned[name].as_string();
               }
               database.appendUnalignedSequences(partition, name, std::move(value));
            }
         }
      }
   } else {  // preprocessor.cpp:255-334: metadata TSV, sequences joined by primary key from FASTA files
      std::vector<std::string> keys;
      {
         LineReader reader(input_path);
         std::string line;
         if (!reader.next(line)) {
            throw PreprocessingException("metadata file " + input_path + " is empty");
         }
         const std::vector<std::string> header = splitTabs(line);
         const auto column_of = [&](const std::string& name) -> size_t {
            const auto found = std::find(header.begin(), header.end(), name);
            if (found == header.end()) {
               // metadata_info.cpp:40-46 (validateFieldsAgainstConfig)
               throw PreprocessingException("The metadata field '" + name + "' which is contained in the database config is not contained in the input.");
            }
            return static_cast<size_t>(found - header.begin());
         };
         const size_t key_column = column_of(schema.primary_key);
         std::vector<size_t> metadata_column_index;
         for (size_t k = 0; k < metadata_writer.size(); ++k) {
            metadata_column_index.push_back(column_of(metadata_writer.name(k)));
         }
         std::vector<std::vector<std::string>> table;  // the metadata file is small next to the sequences: held whole
         while (reader.next(line)) {
            if (line.find_first_not_of(" \t") == std::string::npos) {
               continue;
            }
            table.push_back(splitTabs(line));
         }
         std::vector<uint32_t> order(table.size());
         for (size_t row = 0; row < order.size(); ++row) {
            order[row] = static_cast<uint32_t>(row);
         }
         if (sort_rows) {
            const auto column_values = [&](const std::string& name) {
               const size_t index = column_of(name);
               std::vector<std::string> values;
               values.reserve(table.size());
               for (const auto& fields : table) {
                  values.push_back(index < fields.size() ? fields[index] : "");
               }
               return values;
            };
            order = referenceRowOrder(
               schema.partition_by.has_value() ? column_values(*schema.partition_by) : std::vector<std::string>{},
               schema.date_to_sort_by.has_value() ? column_values(*schema.date_to_sort_by) : std::vector<std::string>{}, column_values(schema.primary_key)
            );
         }
         for (const uint32_t row : order) {
            const std::vector<std::string>& fields = table[row];
            keys.push_back(key_column < fields.size() ? fields[key_column] : "");
            for (size_t k = 0; k < metadata_writer.size(); ++k) {
               const size_t index = metadata_column_index[k];
               metadata_writer.add(k, index < fields.size() ? fields[index] : "");
            }
            metadata_writer.rowDone();
         }
      }
      const auto feed = [&](std::vector<std::pair<std::string, StoreWriter>>& writers, const std::string& prefix) {
         for (auto& [name, writer] : writers) {
            const auto path = findWithCompression(root / (prefix + name + ".fasta"));
            if (!path.has_value()) {
               throw PreprocessingException("no sequence file " + prefix + name + ".fasta[.zst|.xz] in " + directory);
            }
            const auto records = readFasta(*path);
            DatasetSummary counted_before;  // the first of two passes must not count the null sequences twice
            for (int pass = two_pass ? 1 : 2; pass <= 2; ++pass) {
               if (two_pass) {
                  writer.beginPass(pass);
               }
               DatasetSummary& target = pass == 2 ? summary : counted_before;
               for (const std::string& key : keys) {
                  const auto found = records.find(key);
                  writer.add(found == records.end() ? nullptr : &found->second, target);
               }
               writer.flush();
            }
         }
      };
      feed(nuc_writers, setting("nucleotideSequencePrefix", "nuc_"));
      feed(aa_writers, setting("genePrefix", "gene_"));
      // unaligned nucleotide sequences, where a file is present (preprocessor.cpp:355-401)
      for (const auto& [name, store] : partition.nuc_sequences) {
         const auto path = findWithCompression(root / (setting("unalignedNucleotideSequencePrefix", "unaligned_") + name + ".fasta"));
         if (!path.has_value()) {
            continue;
         }
         const auto records = readFasta(*path);
         std::vector<std::optional<std::string>> values;
         values.reserve(keys.size());
         for (const std::string& key : keys) {
            const auto found = records.find(key);
            values.emplace_back(found == records.end() ? std::nullopt : std::optional<std::string>(found->second));
         }
         database.appendUnalignedSequences(partition, name, std::move(values));
      }
   }
   for (auto& [name, writer] : nuc_writers) {
      writer.flush();
   }
   for (auto& [name, writer] : aa_writers) {
      writer.flush();
   }
   metadata_writer.flush();
   database.finalize();
   return summary;
}

}  // namespace silo::preprocessing
