// metadata_columns.cpp — see metadata_columns.h.
#include "metadata_columns.h"

#include <algorithm>
#include <cmath>
#include <cstring>
#include <stdexcept>

#include "database.h"

namespace silo {

namespace common {

Date stringToDate(const std::string& value) {  // date.cpp:22-69
   if (value.empty()) {
      return NULL_DATE;
   }
   const auto split_position = value.find('-', 0);
   if (split_position == std::string::npos) {
      return NULL_DATE;
   }
   const auto split_position2 = value.find('-', split_position + 1);
   if (split_position2 == std::string::npos) {
      return NULL_DATE;
   }
   const std::string year_string = value.substr(0, split_position);
   // the reference passes the END POSITION as the length, so the month string runs into the day; stoi stops at '-'
   const std::string month_string = value.substr(split_position + 1, split_position2);
   const std::string day_string = value.substr(split_position2 + 1);
   try {
      const uint32_t year = static_cast<uint32_t>(std::stoi(year_string));
      const uint32_t month = static_cast<uint32_t>(std::stoi(month_string));
      const uint32_t day = static_cast<uint32_t>(std::stoi(day_string));
      if (month > 12 || month == 0) {
         return NULL_DATE;
      }
      if (day > 31 || day == 0) {
         return NULL_DATE;
      }
      return (year << 16) + (month << 12) + day;
   } catch (const std::invalid_argument&) {
      return NULL_DATE;
   } catch (const std::out_of_range&) {
      return NULL_DATE;
   }
}

std::optional<std::string> dateToString(Date date) {  // date.cpp:71-86
   if (date == 0) {
      return std::nullopt;
   }
   const uint32_t year = date >> 16;
   const uint32_t month = (date >> 12) & 0xF;
   const uint32_t day = date & 0xFFF;
   char buffer[40];
   snprintf(buffer, sizeof(buffer), "%04u-%02u-%02u", year, month, day);
   return std::string(buffer);
}

}  // namespace common

namespace config {

std::optional<ColumnType> columnTypeFromConfig(const std::string& type, bool generate_index) {  // database_config.cpp:24-47,158-189
   if (type == "string") {
      return generate_index ? ColumnType::INDEXED_STRING : ColumnType::STRING;
   }
   if (type == "pango_lineage") {
      return ColumnType::INDEXED_PANGOLINEAGE;
   }
   if (type == "date") {
      return ColumnType::DATE;
   }
   if (type == "int") {
      return ColumnType::INT;
   }
   if (type == "float") {
      return ColumnType::FLOAT;
   }
   if (type == "insertion") {
      return ColumnType::NUC_INSERTION;
   }
   if (type == "aaInsertion") {
      return ColumnType::AA_INSERTION;
   }
   return std::nullopt;
}

std::optional<storage::ColumnMetadata> DatabaseConfig::getMetadata(const std::string& name) const {
   for (const auto& entry : metadata) {
      if (entry.name == name) {
         return entry;
      }
   }
   return std::nullopt;
}

}  // namespace config

namespace storage::column {

using config::ColumnType;

MetadataColumnPartition::MetadataColumnPartition(ColumnType type, bool is_sorted, const PangoLineageAliasLookup* alias_key)
    : type(type), is_sorted(is_sorted), alias_key_(alias_key) {}

MetadataColumnPartition::~MetadataColumnPartition() {
   silo_gpu_free(device_values_);
   silo_gpu_free(numeric_groups_.device_ids);
}

bool MetadataColumnPartition::isStringLike() const {
   return type != ColumnType::INT && type != ColumnType::FLOAT && type != ColumnType::DATE;
}

size_t MetadataColumnPartition::numRows() const {
   if (type == ColumnType::INT) {
      return ints.size();
   }
   if (type == ColumnType::FLOAT) {
      return floats.size();
   }
   return words.size();
}

void MetadataColumnPartition::reserve(size_t row_count) {
   if (type == ColumnType::INT) {
      ints.reserve(row_count);
   } else if (type == ColumnType::FLOAT) {
      floats.reserve(row_count);
   } else {
      words.reserve(row_count);
   }
}

void MetadataColumnPartition::insert(const std::string& text) {
   switch (type) {
      case ColumnType::INT:  // int_column.cpp:17-24
         try {
            ints.push_back(text.empty() ? INT32_MIN : std::stoi(text));
         } catch (const std::logic_error&) {
            throw std::runtime_error("Wrong format for Integer: '" + text + "'");
         }
         return;
      case ColumnType::FLOAT:  // float_column.cpp:16-24
         try {
            floats.push_back(text.empty() ? std::nan("") : std::stod(text));
         } catch (const std::logic_error&) {
            throw std::runtime_error("Bad format for double value: '" + text + "'");
         }
         return;
      case ColumnType::DATE:  // date_column.cpp:15-17 after stringToDate
         words.push_back(common::stringToDate(text));
         return;
      default: break;
   }
   std::string value = text;
   if (type == ColumnType::INDEXED_PANGOLINEAGE && alias_key_ != nullptr) {
      // what lookupAliasedValue returns for the row: the unaliased lineage, aliased again (pango_lineage_column.cpp:21-38,86-88)
      value = alias_key_->aliasPangoLineage(alias_key_->unaliasPangoLineage(text));
   }
   auto found = lookup_.find(value);
   if (found == lookup_.end()) {
      found = lookup_.emplace(value, static_cast<uint32_t>(dictionary.size())).first;
      dictionary.push_back(value);
   }
   words.push_back(found->second);
}

std::optional<uint32_t> MetadataColumnPartition::lookupId(const std::string& value) const {
   const auto found = lookup_.find(value);
   if (found == lookup_.end()) {
      return std::nullopt;
   }
   return found->second;
}

int MetadataColumnPartition::deviceValueType() const {
   if (type == ColumnType::INT) {
      return SILO_GPU_VALUE_I32;
   }
   if (type == ColumnType::FLOAT) {
      return SILO_GPU_VALUE_F64;
   }
   return SILO_GPU_VALUE_U32;
}

void MetadataColumnPartition::finalize() {
   silo_gpu_free(device_values_);
   device_values_ = nullptr;
   if (numRows() == 0) {
      return;
   }
   const void* host = type == ColumnType::INT     ? static_cast<const void*>(ints.data())
                      : type == ColumnType::FLOAT ? static_cast<const void*>(floats.data())
                                                  : static_cast<const void*>(words.data());
   checkGpu(silo_gpu_upload_column(host, numRows(), deviceValueType(), &device_values_), "silo_gpu_upload_column");
   const std::lock_guard<std::mutex> lock(groups_mutex_);
   silo_gpu_free(numeric_groups_.device_ids);
   numeric_groups_ = NumericGroups{};
}

namespace {

/// Total order used to rank raw doubles for grouping: bitwise distinct values stay distinct (the reference keys
/// tuples by their bytes, tuple.cpp:389-391), NaNs after every number.
bool doubleBitsLess(double a, double b) {
   const bool a_nan = std::isnan(a);
   const bool b_nan = std::isnan(b);
   if (a_nan || b_nan) {
      if (a_nan && b_nan) {
         uint64_t a_bits = 0, b_bits = 0;
         std::memcpy(&a_bits, &a, sizeof(a));
         std::memcpy(&b_bits, &b, sizeof(b));
         return a_bits < b_bits;
      }
      return b_nan;
   }
   if (a == b) {
      return std::signbit(a) && !std::signbit(b);  // -0.0 before +0.0
   }
   return a < b;
}

template <typename T, typename Less>
uint32_t* rankValues(const std::vector<T>& values, std::vector<T>& distinct, Less less) {
   distinct = values;
   std::sort(distinct.begin(), distinct.end(), less);
   distinct.erase(
      std::unique(distinct.begin(), distinct.end(), [&](const T& a, const T& b) { return !less(a, b) && !less(b, a); }), distinct.end()
   );
   std::vector<uint32_t> ids(values.size());
   for (size_t row = 0; row < values.size(); ++row) {
      ids[row] = static_cast<uint32_t>(std::lower_bound(distinct.begin(), distinct.end(), values[row], less) - distinct.begin());
   }
   void* device = nullptr;
   checkGpu(silo_gpu_upload_column(ids.data(), ids.size(), SILO_GPU_VALUE_U32, &device), "silo_gpu_upload_column");
   return static_cast<uint32_t*>(device);
}

}  // namespace

MetadataColumnPartition::Groups MetadataColumnPartition::groups() const {
   if (isStringLike()) {
      return {static_cast<const uint32_t*>(device_values_), static_cast<uint32_t>(dictionary.size())};
   }
   const std::lock_guard<std::mutex> lock(groups_mutex_);
   NumericGroups& groups = numeric_groups_;
   if (!groups.ready && numRows() != 0) {
      if (type == ColumnType::INT) {
         groups.device_ids = rankValues(ints, groups.ints, std::less<int32_t>());
         groups.cardinality = static_cast<uint32_t>(groups.ints.size());
      } else if (type == ColumnType::FLOAT) {
         groups.device_ids = rankValues(floats, groups.floats, doubleBitsLess);
         groups.cardinality = static_cast<uint32_t>(groups.floats.size());
      } else {
         groups.device_ids = rankValues(words, groups.words, std::less<uint32_t>());
         groups.cardinality = static_cast<uint32_t>(groups.words.size());
      }
      groups.ready = true;
   }
   return {groups.device_ids, groups.cardinality};
}

namespace {

JsonValue jsonOfInt(int32_t value) {
   if (value == INT32_MIN) {
      return std::nullopt;
   }
   return value;
}
JsonValue jsonOfFloat(double value) {
   if (std::isnan(value)) {
      return std::nullopt;
   }
   return value;
}
JsonValue jsonOfDate(common::Date value) {
   const auto text = common::dateToString(value);
   if (!text.has_value()) {
      return std::nullopt;
   }
   return *text;
}
JsonValue jsonOfString(const std::string& value) {
   if (value.empty()) {
      return std::nullopt;
   }
   return value;
}

template <typename T>
void appendRaw(std::string& key, const T& value) {
   key.append(reinterpret_cast<const char*>(&value), sizeof(value));
}

}  // namespace

JsonValue MetadataColumnPartition::jsonOfRow(uint32_t row) const {
   switch (type) {
      case ColumnType::INT: return jsonOfInt(ints.at(row));
      case ColumnType::FLOAT: return jsonOfFloat(floats.at(row));
      case ColumnType::DATE: return jsonOfDate(words.at(row));
      default: return jsonOfString(dictionary.at(words.at(row)));
   }
}

JsonValue MetadataColumnPartition::jsonOfGroup(uint32_t group) const {
   switch (type) {
      case ColumnType::INT: return jsonOfInt(numeric_groups_.ints.at(group));
      case ColumnType::FLOAT: return jsonOfFloat(numeric_groups_.floats.at(group));
      case ColumnType::DATE: return jsonOfDate(numeric_groups_.words.at(group));
      default: return jsonOfString(dictionary.at(group));
   }
}

void MetadataColumnPartition::appendKeyOfRow(uint32_t row, std::string& key) const {
   switch (type) {
      case ColumnType::INT: appendRaw(key, ints.at(row)); return;
      case ColumnType::FLOAT: appendRaw(key, floats.at(row)); return;
      case ColumnType::DATE: appendRaw(key, words.at(row)); return;
      default:
         key += dictionary.at(words.at(row));
         key.push_back('\0');
   }
}

void MetadataColumnPartition::appendKeyOfGroup(uint32_t group, std::string& key) const {
   switch (type) {
      case ColumnType::INT: appendRaw(key, numeric_groups_.ints.at(group)); return;
      case ColumnType::FLOAT: appendRaw(key, numeric_groups_.floats.at(group)); return;
      case ColumnType::DATE: appendRaw(key, numeric_groups_.words.at(group)); return;
      default:
         key += dictionary.at(group);
         key.push_back('\0');
   }
}

int MetadataColumnPartition::compareRows(uint32_t row, const MetadataColumnPartition& other, uint32_t other_row) const {
   switch (type) {
      case ColumnType::INT: {
         const int32_t a = ints.at(row), b = other.ints.at(other_row);
         return a < b ? -1 : (a > b ? 1 : 0);
      }
      case ColumnType::FLOAT: {  // compareDouble, tuple.cpp:162-182: NaN after every number
         const double a = floats.at(row), b = other.floats.at(other_row);
         if (a < b) {
            return -1;
         }
         if (a > b) {
            return 1;
         }
         if (a == b) {
            return 0;
         }
         if (std::isnan(b)) {
            return std::isnan(a) ? 0 : -1;
         }
         return 1;
      }
      case ColumnType::DATE: {
         const uint32_t a = words.at(row), b = other.words.at(other_row);
         return a < b ? -1 : (a > b ? 1 : 0);
      }
      default: {
         const int compared = dictionary.at(words.at(row)).compare(other.dictionary.at(other.words.at(other_row)));
         return compared < 0 ? -1 : (compared > 0 ? 1 : 0);
      }
   }
}

// ---- InsertionColumnPartition ---------------------------------------------------------------------------
InsertionColumnPartition::InsertionColumnPartition(std::optional<std::string> default_sequence_name)
    : default_sequence_name_(std::move(default_sequence_name)) {}

InsertionColumnPartition::~InsertionColumnPartition() {
   for (auto& [name, index] : indexes_) {
      silo_gpu_free(index.device_rows);
      silo_gpu_free(index.device_ids);
   }
}

namespace {

std::vector<std::string> splitBy(const std::string& value, char delimiter) {  // string_utils.cpp:8-23
   std::vector<std::string> splits;
   size_t begin = 0;
   while (true) {
      const auto next = value.find(delimiter, begin);
      splits.push_back(value.substr(begin, next == std::string::npos ? std::string::npos : next - begin));
      if (next == std::string::npos) {
         return splits;
      }
      begin = next + 1;
   }
}

uint32_t parsePosition(const std::string& text, const std::string& whole) {  // boost::lexical_cast<uint32_t>
   if (text.empty() || text.find_first_not_of("0123456789") != std::string::npos || text.size() > 10) {
      throw std::runtime_error("Failed to parse insertion due to invalid format: " + whole);
   }
   const unsigned long long value = std::stoull(text);
   if (value > UINT32_MAX) {
      throw std::runtime_error("Failed to parse insertion due to invalid format: " + whole);
   }
   return static_cast<uint32_t>(value);
}

}  // namespace

std::string InsertionColumnPartition::insert(const std::string& value, uint32_t row) {
   if (value.empty()) {
      return "";
   }
   std::string standardized_value;
   for (const std::string& insertion_entry : splitBy(value, ',')) {
      std::vector<std::string> parts = splitBy(insertion_entry, ':');  // parseInsertion, insertion_column.cpp:31-65
      for (std::string& part : parts) {
         part.erase(std::remove(part.begin(), part.end(), '"'), part.end());
      }
      std::string sequence_name;
      uint32_t position = 0;
      std::string insertion;
      if (parts.size() == 2 && default_sequence_name_.has_value()) {
         sequence_name = *default_sequence_name_;
         position = parsePosition(parts[0], insertion_entry);
         insertion = parts[1];
      } else if (parts.size() == 3) {
         sequence_name = parts[0];
         position = parsePosition(parts[1], insertion_entry);
         insertion = parts[2];
      } else {
         throw std::runtime_error("Failed to parse insertion due to invalid format: " + insertion_entry);
      }
      SequenceIndex& index = indexes_[sequence_name];
      const auto key = std::make_pair(position, insertion);
      auto found = index.lookup.find(key);
      if (found == index.lookup.end()) {
         const auto id = static_cast<uint32_t>(index.insertions.size());
         found = index.lookup.emplace(key, id).first;
         index.positions.push_back(position);
         index.insertions.push_back(insertion);
         index.ids_at_position[position].push_back(id);
      }
      index.pair_rows.push_back(row);
      index.pair_ids.push_back(found->second);
      if (!standardized_value.empty()) {
         standardized_value += ",";
      }
      if (default_sequence_name_.has_value() && *default_sequence_name_ == sequence_name) {
         standardized_value += std::to_string(position) + ":" + insertion;
      } else {
         standardized_value += sequence_name + ":" + std::to_string(position) + ":" + insertion;
      }
   }
   return standardized_value;
}

void InsertionColumnPartition::finalize() {
   for (auto& [name, index] : indexes_) {
      silo_gpu_free(index.device_rows);
      silo_gpu_free(index.device_ids);
      index.device_rows = nullptr;
      index.device_ids = nullptr;
      if (index.pair_rows.empty()) {
         continue;
      }
      void* device = nullptr;
      checkGpu(silo_gpu_upload_column(index.pair_rows.data(), index.pair_rows.size(), SILO_GPU_VALUE_U32, &device), "silo_gpu_upload_column");
      index.device_rows = static_cast<uint32_t*>(device);
      checkGpu(silo_gpu_upload_column(index.pair_ids.data(), index.pair_ids.size(), SILO_GPU_VALUE_U32, &device), "silo_gpu_upload_column");
      index.device_ids = static_cast<uint32_t*>(device);
   }
}

}  // namespace storage::column

}  // namespace silo
