// metadata_columns.h — metadata columns of a partition and the schema that names them.
//
//   config::ColumnType / DatabaseConfig     include/silo/config/database_config.h:14-62, src/.../database_config.cpp:158-189
//   common::stringToDate / dateToString     src/silo/common/date.cpp:22-86
//   storage::column::MetadataColumnPartition  string_column.cpp, indexed_string_column.cpp, int_column.cpp,
//                                           float_column.cpp, date_column.cpp, insertion_column.cpp (text only),
//                                           pango_lineage_column.cpp:86-92 (lookupAliasedValue)
//
// One class serves every column type: the raw values of a partition live on the host (Details, result rendering)
// and in HBM (predicates through k_bitset_from_compare, group-by through k_group_count).  String-like columns are
// dictionary encoded — the reference does that only for "indexed" columns; with a device compare kernel the
// encoding is what makes StringEquals a 4-byte-per-row stream for the plain ones as well.
#pragma once

#include <cstdint>
#include <map>
#include <memory>
#include <mutex>
#include <optional>
#include <string>
#include <unordered_map>
#include <variant>
#include <vector>

namespace silo {

class PangoLineageAliasLookup;

namespace common {
using Date = uint32_t;  // year << 16 | month << 12 | day, 0 = NULL (date.h)
constexpr Date NULL_DATE = 0;
Date stringToDate(const std::string& value);
std::optional<std::string> dateToString(Date date);
}  // namespace common

namespace config {
enum class ColumnType { STRING, INDEXED_STRING, INDEXED_PANGOLINEAGE, DATE, INT, FLOAT, NUC_INSERTION, AA_INSERTION };
/// "string" (+ generateIndex), "pango_lineage", "date", "int", "float", "insertion", "aaInsertion" of database_config.yaml.
std::optional<ColumnType> columnTypeFromConfig(const std::string& type, bool generate_index);
}  // namespace config

namespace storage {
struct ColumnMetadata {
   std::string name;
   config::ColumnType type;
};
}  // namespace storage

namespace config {
struct DatabaseConfig {  // database_config.h: default_nucleotide_sequence, schema.{metadata (file order), primary_key, date_to_sort_by}
   std::string default_nucleotide_sequence = "main";
   std::vector<storage::ColumnMetadata> metadata;
   std::string primary_key;
   std::optional<std::string> date_to_sort_by;
   [[nodiscard]] std::optional<storage::ColumnMetadata> getMetadata(const std::string& name) const;
};
}  // namespace config

/// A field of a result row: query_result.h:14-20.
using JsonValue = std::optional<std::variant<std::string, int32_t, double>>;

namespace storage::column {

class MetadataColumnPartition {
  public:
   /// `alias_key` is needed for INDEXED_PANGOLINEAGE only (values are rendered re-aliased, pango_lineage_column.cpp:86-88).
   MetadataColumnPartition(config::ColumnType type, bool is_sorted, const PangoLineageAliasLookup* alias_key);
   ~MetadataColumnPartition();
   MetadataColumnPartition(const MetadataColumnPartition&) = delete;
   MetadataColumnPartition& operator=(const MetadataColumnPartition&) = delete;

   /// One row from its text form ("" = NULL), as the reference's column inserts parse it.
   void insert(const std::string& text);
   void reserve(size_t row_count);
   /// Uploads the raw values.
   void finalize();
   [[nodiscard]] size_t numRows() const;

   const config::ColumnType type;
   const bool is_sorted;  // DATE only: the column named by dateToSortBy (date_column.cpp:9-13)

   [[nodiscard]] bool isStringLike() const;
   /// Dictionary id of a value of a string-like column.
   [[nodiscard]] std::optional<uint32_t> lookupId(const std::string& value) const;

   /// Raw values in HBM: int32 (INT), double (FLOAT), uint32 (DATE, dictionary ids); SILO_GPU_VALUE_* type.
   [[nodiscard]] const void* deviceValues() const { return device_values_; }
   [[nodiscard]] int deviceValueType() const;

   /// Dense ids for group-by: the dictionary ids of a string-like column; for INT / FLOAT / DATE the rank of the
   /// raw value among the distinct values of the partition (built and uploaded on first use).
   struct Groups {
      const uint32_t* device_ids = nullptr;
      uint32_t cardinality = 0;
   };
   [[nodiscard]] Groups groups() const;

   // ---- rendering and ordering (tuple.cpp) -------------------------------------------------------
   [[nodiscard]] JsonValue jsonOfRow(uint32_t row) const;    // tupleFieldToValueType :82-160
   [[nodiscard]] JsonValue jsonOfGroup(uint32_t group) const;
   /// The raw bytes a Tuple holds for this field (assignTupleField :29-80), strings by content so that tuples of
   /// different partitions compare equal when their values do.
   void appendKeyOfRow(uint32_t row, std::string& key) const;
   void appendKeyOfGroup(uint32_t group, std::string& key) const;
   /// compareTupleFields :184-290: <0, 0, >0.
   [[nodiscard]] int compareRows(uint32_t row, const MetadataColumnPartition& other, uint32_t other_row) const;

   // raw host values; exactly one of the three is in use
   std::vector<int32_t> ints;          // INT, NULL = INT32_MIN
   std::vector<double> floats;         // FLOAT, NULL = NaN
   std::vector<uint32_t> words;        // DATE values, or dictionary ids of string-like columns
   std::vector<std::string> dictionary;  // string-like: id -> rendered value ("" = NULL)

  private:
   const PangoLineageAliasLookup* alias_key_;
   std::unordered_map<std::string, uint32_t> lookup_;
   void* device_values_ = nullptr;

   struct NumericGroups {
      std::vector<int32_t> ints;
      std::vector<double> floats;
      std::vector<uint32_t> words;
      uint32_t* device_ids = nullptr;
      uint32_t cardinality = 0;
      bool ready = false;
   };
   mutable std::mutex groups_mutex_;
   mutable NumericGroups numeric_groups_;
};

/// The insertion index of one insertion column of one partition (insertion_column.cpp, insertion_index.cpp).
/// Per sequence name: the distinct (position, insertion) values and their occurrences as (row, id) pairs, on the host
/// (pattern search over the distinct insertions of a position) and in HBM (k_bitset_from_pairs / k_count_pairs).
class InsertionColumnPartition {
  public:
   explicit InsertionColumnPartition(std::optional<std::string> default_sequence_name);
   ~InsertionColumnPartition();
   InsertionColumnPartition(const InsertionColumnPartition&) = delete;
   InsertionColumnPartition& operator=(const InsertionColumnPartition&) = delete;

   /// Indexes the insertions of row `row` ("" = none) and returns the standardised text the column holds for it
   /// (insertion_column.cpp:76-113).  Throws std::runtime_error for an entry that is not [sequence:]position:insertion.
   std::string insert(const std::string& value, uint32_t row);
   void finalize();  // uploads the pairs

   struct SequenceIndex {
      std::vector<uint32_t> positions;       // per distinct insertion id
      std::vector<std::string> insertions;   // per distinct insertion id
      std::map<uint32_t, std::vector<uint32_t>> ids_at_position;
      std::map<std::pair<uint32_t, std::string>, uint32_t> lookup;
      std::vector<uint32_t> pair_rows;
      std::vector<uint32_t> pair_ids;
      uint32_t* device_rows = nullptr;
      uint32_t* device_ids = nullptr;
   };
   [[nodiscard]] const std::map<std::string, SequenceIndex>& getInsertionIndexes() const { return indexes_; }

  private:
   std::optional<std::string> default_sequence_name_;
   std::map<std::string, SequenceIndex> indexes_;
};

}  // namespace storage::column

}  // namespace silo
