// host_logic.cpp — test hooks into the pure host functions of libsilo_engine.so (no device is touched):
// date parsing, lineage (un)aliasing, insertion standardisation.  Built by silo_amd/build.py for the CPU tests.
#include <cstring>
#include <string>

#include <chrono>

#include "../../lapis-silo_amd/csrc/layout_choice.h"
#include "database.h"
#include "dataset_loader.h"
#include "query_engine.h"

namespace {
int copyOut(const std::string& text, char* out, size_t capacity) {
   if (text.size() + 1 > capacity) {
      return -2;
   }
   std::memcpy(out, text.c_str(), text.size() + 1);
   return static_cast<int>(text.size());
}
}  // namespace

extern "C" {

/// chooseLayouts of csrc/layout_choice.h on totals[positions][n_scan] for rows of `sequences` sequences (row bytes as the
/// device library pads them); code_map_out[positions][8], escape_count_out[positions][n_scan].  allow_one_hot: silo_gpu_layout::OneHotMode.
void t_choose_layouts(
   const uint32_t* totals, uint32_t n_scan, uint32_t n_bits, uint32_t positions, uint64_t sequences, int allow_one_hot, uint64_t key_cost,
   uint8_t* code_map_out, uint32_t* escape_count_out
) {
   const uint64_t row_words = ((sequences + 63) / 64 + 31) / 32 * 32;
   std::vector<uint32_t> counts(totals, totals + static_cast<size_t>(positions) * n_scan);
   std::vector<uint8_t> code_map;
   std::vector<uint32_t> escapes;
   silo_gpu_layout::chooseLayouts(
      counts, n_scan, n_bits, positions, row_words * 8, allow_one_hot, key_cost != 0 ? key_cost : silo_gpu_layout::KEY_COST_BYTES, code_map, escapes
   );
   std::memcpy(code_map_out, code_map.data(), code_map.size());
   std::memcpy(escape_count_out, escapes.data(), escapes.size() * sizeof(uint32_t));
}

/// referenceRowOrder of host/dataset_loader.cpp over n rows given as three arrays of C strings (partition_keys / dates may be null).
void t_reference_row_order(const char* const* partition_keys, const char* const* dates, const char* const* primary_keys, uint32_t n, uint32_t* order_out) {
   const auto strings = [n](const char* const* values) {
      std::vector<std::string> out;
      for (uint32_t k = 0; values != nullptr && k < n; ++k) {
         out.emplace_back(values[k]);
      }
      return out;
   };
   const std::vector<uint32_t> order = silo::preprocessing::referenceRowOrder(strings(partition_keys), strings(dates), strings(primary_keys));
   std::memcpy(order_out, order.data(), order.size() * sizeof(uint32_t));
}

uint32_t t_string_to_date(const char* text) {
   return silo::common::stringToDate(text);
}

int t_date_to_string(uint32_t date, char* out, size_t capacity) {
   const auto text = silo::common::dateToString(date);
   return text.has_value() ? copyOut(*text, out, capacity) : -1;
}

/// mode 0: unalias, 1: alias (of an unaliased lineage), 2: alias(unalias(lineage)) — what a lineage column renders
int t_lineage(const char* alias_json, const char* lineage, int mode, char* out, size_t capacity) {
   try {
      const auto lookup = silo::PangoLineageAliasLookup::fromJson(silo::json::parse(alias_json));
      std::string value = lineage;
      if (mode == 0 || mode == 2) {
         value = lookup.unaliasPangoLineage(value);
      }
      if (mode == 1 || mode == 2) {
         value = lookup.aliasPangoLineage(value);
      }
      return copyOut(value, out, capacity);
   } catch (const std::exception&) {
      return -1;
   }
}

/// Standardised text of an insertion column value (default_sequence may be NULL); -1 for a malformed value.
int t_insertion_standardise(const char* default_sequence, const char* value, char* out, size_t capacity) {
   try {
      silo::storage::column::InsertionColumnPartition column(
         default_sequence != nullptr ? std::optional<std::string>(default_sequence) : std::nullopt
      );
      return copyOut(column.insert(value, 0), out, capacity);
   } catch (const std::exception&) {
      return -1;
   }
}

/// What the loader understands of a database_config.yaml, as JSON (return >= 0), or the validation error (return -1).
int t_describe_database_config(const char* path, int validate, char* out, size_t capacity) {
   try {
      return copyOut(silo::preprocessing::describeDatabaseConfig(path, validate != 0), out, capacity);
   } catch (const std::exception& error) {
      (void)copyOut(error.what(), out, capacity);
      return -1;
   }
}

/// The records the loader reads from a FASTA file as JSON [[key, genome], ...] (return >= 0), or the error (return -1).
int t_describe_fasta(const char* path, char* out, size_t capacity) {
   try {
      return copyOut(silo::preprocessing::describeFasta(path), out, capacity);
   } catch (const std::exception& error) {
      (void)copyOut(error.what(), out, capacity);
      return -1;
   }
}

/// Average microseconds to parse `query_json` into a Query (JSON -> Expression tree + Action), or -1 if it is invalid.
double t_parse_query_us(const char* query_json, int repetitions) {
   try {
      const std::string text = query_json;
      const auto start = std::chrono::steady_clock::now();
      for (int i = 0; i < repetitions; ++i) {
         const silo::query_engine::Query query(text);
         if (query.filter == nullptr) {
            return -1;
         }
      }
      return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - start).count() / repetitions;
   } catch (const std::exception&) {
      return -1;
   }
}

}  // extern "C"
