// store_internal.h — what the translation units of libsilo_gpu.so share about a store (not part of the C ABI): the device-side
// description of a sequence store, its host-side state, the store itself, the wave-level helpers and a few constants.
//   silo_gpu_runtime.hip   errors, tuning knobs, memory / stream / event wrappers, the stream-read probe
//   silo_gpu_store.hip     store lifetime, the build kernels (transpose, generator), finalize: runs of the missing symbol, layout
//   silo_gpu_scan.hip      K1 the Mutations scan (plane rows, escape keys, derived symbols), K4 row selection, row slots
//   silo_gpu_filter.hip    K2 / K3 / K3b filter evaluation, count slots, bitsets, planes of single symbols, FastaAligned
//   silo_gpu_import.hip    import of the reference's roaring payloads
//   silo_gpu_columns.hip, silo_gpu_comm.hip, silo_gpu_sort.hip   metadata columns, RCCL, rocPRIM sorts
#pragma once
#include <hip/hip_runtime.h>

#include <stdint.h>

#include <atomic>
#include <memory>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/silo_gpu.h"
#include "bitprog.h"
#include "internal.h"
#include "layout_choice.h"

// process-wide tuning knobs (silo_gpu_tune; defined in silo_gpu_runtime.hip): for probes and tests
extern std::atomic<int> g_tune_rows_per_block;
extern std::atomic<int> g_tune_scan_variant;
extern std::atomic<int> g_tune_eval_leaf_batch;
extern std::atomic<int> g_tune_compact_index;   // < 0: finalize keeps the build-time identity planes; 2 / 3: see SILO_GPU_TUNE_COMPACT_INDEX
extern std::atomic<int> g_tune_side_stream;     // the side passes of a scan: see forkSidePasses (silo_gpu_scan.hip)
extern std::atomic<int> g_tune_scan_timing;     // 1: HIP events around every launch of a scan (silo_gpu_scan_timings)
extern std::atomic<int> g_tune_missing_runs;    // < 0: finalize keeps the plane of the missing symbol instead of turning it into runs
extern std::atomic<int> g_tune_key_cost;        // > 0: what an escape key costs in plane bytes in the layout choice (default KEY_COST_BYTES)
extern std::atomic<int> g_tune_launch_cost;     // KiB of plane bytes a further kind of plane-scan launch costs in the layout choice: 0 = default, < 0 = nothing
extern std::atomic<int> g_tune_sparse_divisor;  // 0 = default (row_words / 16 filter sectors with a set bit), < 0 = sparse-filter path off

namespace silo_gpu_detail {

inline int fail(int code, const std::string& message) {
   return silo_gpu_internal_fail(code, message);
}

#define HIP_TRY(expr)                                                                          \
   do {                                                                                        \
      hipError_t err_ = (expr);                                                                \
      if (err_ != hipSuccess) {                                                                \
         (void)hipGetLastError(); /* clear the sticky error so later launch checks start clean */ \
         return fail(                                                                          \
            err_ == hipErrorOutOfMemory ? SILO_GPU_ERR_OUT_OF_MEMORY : SILO_GPU_ERR_HIP,       \
            std::string(#expr) + ": " + hipGetErrorString(err_)                                \
         );                                                                                    \
      }                                                                                        \
   } while (0)

constexpr uint32_t ROW_ALIGN_WORDS = 32;  // 256-byte rows

// ------------------------------------------------------------------------------------------------
// alphabets (host side tables; ids = reference enum values)
// nucleotide_symbols.cpp:46-85  /  aa_symbols.cpp:62-117
// ------------------------------------------------------------------------------------------------
inline void fillCharTable(uint32_t alphabet, uint8_t table[256]) {
   memset(table, SILO_GPU_SYMBOL_NONE, 256);
   if (alphabet == SILO_GPU_ALPHABET_NUCLEOTIDE) {
      const char* symbols = "-ACGTRYSWKMBDHVN";
      for (int i = 0; i < 16; ++i) {
         table[static_cast<uint8_t>(symbols[i])] = static_cast<uint8_t>(i);
      }
      table[static_cast<uint8_t>('.')] = 0;  // '.' -> GAP   (nucleotide_symbols.cpp:48-50)
      table[static_cast<uint8_t>('U')] = 4;  // 'U' -> T     (nucleotide_symbols.cpp:58-60)
   } else {
      const char* symbols = "-ACDEFGHIKLMNPQRSTVWYBZ*X";  // enum order, STOP = 23, X = 24
      for (int i = 0; i < 25; ++i) {
         table[static_cast<uint8_t>(symbols[i])] = static_cast<uint8_t>(i);
      }
   }
}

inline uint32_t alphabetSize(uint32_t alphabet) {
   return alphabet == SILO_GPU_ALPHABET_NUCLEOTIDE ? SILO_GPU_NUC_SYMBOLS : SILO_GPU_AA_SYMBOLS;
}
inline uint32_t missingSymbol(uint32_t alphabet) {
   return alphabet == SILO_GPU_ALPHABET_NUCLEOTIDE ? 15u : 24u;  // N / X
}


// ------------------------------------------------------------------------------------------------
// device-side description of one sequence store (passed to kernels by value)
// ------------------------------------------------------------------------------------------------
enum : uint8_t { PLANE_SPARSE = 0, PLANE_SCAN = 1, PLANE_EXTRA = 2, PLANE_RUNS = 3 };
enum : uint32_t { BUILD_PLANES = 0, BUILD_COUNT = 1, BUILD_ENCODE = 2 };

// PLANE_RUNS: after finalize the missing symbol (N / X: amplicon drop-outs, unsequenced ends — long runs of a row, 0.5 % of
// the cells but a plane per position, half of a finished nucleotide store) is kept as the sorted list of its runs — positions
// [start, end) of one sequence — instead: the reference keeps it row-wise too (missing_symbol_bitmaps,
// sequence_store.cpp:153-190).  A position's plane is materialised from the runs when a filter leaf asks for it.

// Layout of a position in the adaptive planes (code_map[p][0], layout_choice.h): the number of plane rows, and whether they
// are identity code planes (code = index of the valid mutation symbol + 1, no escapes: the position keeps its full planes) ...
using silo_gpu_layout::LAYOUT_IDENTITY;
// ... or ONE-HOT rows: the low bits give k = 1..3 rows, row j holds exactly the rows of the position's j-th most frequent
// valid symbol (code_map[p][1 + j]); every other valid symbol of a row is an escape key.  Rows of one-hot positions need no
// joint decoding — each is one AND + popcount under the filter — so positions with different k form ONE run for the scan.
using silo_gpu_layout::LAYOUT_ONE_HOT;
// ... of which the position's most numerous symbol (code_map[p][IMPLICIT_SLOT]) may be IMPLICIT: no row, no keys — its count
// under a filter is what is left of the filter's rows once the rows without a valid symbol (runs of the missing symbol,
// ambiguity codes) and the other valid symbols are taken away (k_finish_scan), its plane the complement of everything else.
using silo_gpu_layout::LAYOUT_IMPLICIT;
using silo_gpu_layout::LAYOUT_ROWS_MASK;
using silo_gpu_layout::IMPLICIT_SLOT;
using silo_gpu_layout::CODE_MAP_STRIDE;  // bytes of code_map per position: [0] = layout, [c] = scan symbol of code c (1..7), 0xFF = unused

struct SeqStoreDev {
   // BUILD-TIME bit-sliced planes [P][n_bits][Wp]: bit b of the CODE of every row's symbol at the position, where the code
   // of the k-th valid mutation symbol is k + 1 and 0 stands for "none of them" (missing, ambiguity code, row padding).
   // n_bits = 3 for the 5 nucleotide symbols, 5 for the 22 amino-acid symbols.  append / generate write here;
   // finalize re-encodes them into the adaptive planes below and frees them (scan == nullptr from then on) unless the
   // store keeps them as they are (short rows, compact layouts switched off): then planes == scan.
   uint64_t* scan;
   uint64_t* extra;  // [n_extra][P][Wp]; nullptr once the missing symbol's plane has become runs (kind PLANE_RUNS)
   const uint64_t* missing_run_keys;  // sequence << 32 | start, ascending
   const uint32_t* missing_run_ends;  // the run's end (exclusive)
   uint32_t n_missing_runs;
   // Two-pass build (silo_gpu_store_build_pass): BUILD_COUNT only counts the valid symbols per position (enc_counts
   // [P][n_scan]); BUILD_ENCODE writes every row straight into its position's adaptive layout, chosen from those counts —
   // no build-time planes at all.
   uint32_t build_mode;
   uint32_t* enc_counts;
   const uint8_t* enc_code_map;
   const uint32_t* enc_row_of;
   uint64_t* enc_planes;
   const uint32_t* enc_first;   // [P * n_scan + 1] first escape key of a (position, symbol)
   uint32_t* enc_cursor;        // [P * n_scan] keys written so far
   uint64_t* enc_escapes;
   // ... and the missing symbol goes straight to its runs (no plane): counted in the first pass, written in the second
   uint32_t runs_at_build;
   unsigned long long* enc_run_count;  // the count, then the cursor
   uint64_t* enc_run_keys;
   uint32_t* enc_run_ends;
   unsigned long long enc_run_capacity;
   // ADAPTIVE code planes, what every consumer reads after finalize.  Position p owns plane rows
   // [row_of[p], row_of[p + 1]) of `planes`: B = 2 or 3 planes carrying the codes 1..2^B-1 of the position's most frequent
   // valid symbols (code_map), every other valid symbol of a row listed in `escapes`; or the n_bits identity planes.
   // While a store is being built row_of / code_map are null: position p then sits at row p * n_bits with identity codes.
   const uint64_t* planes;
   const uint32_t* row_of;         // [P + 1]
   const uint8_t* code_map;        // [P][CODE_MAP_STRIDE]
   const uint64_t* escapes;        // position << 37 | scan symbol index << 32 | sequence, ascending
   const uint32_t* escape_first;   // [P + 1] first key of a position
   uint32_t positions;
   uint32_t n_symbols;  // alphabet size
   uint32_t n_scan;
   uint32_t n_bits;
   uint32_t n_extra;
   uint32_t row_words;  // Wp
   uint32_t missing_symbol;
   uint8_t kind[SILO_GPU_MAX_SYMBOLS];
   uint8_t index[SILO_GPU_MAX_SYMBOLS];
};

/// One-hot plane of a symbol that has one: the extra symbols.  Valid mutation symbols live in the code planes
/// (decodeCodeWord / silo_gpu_store_sparse_plane materialise their one-hot plane on demand).
__host__ __device__ inline uint64_t* planePtr(const SeqStoreDev& s, uint32_t position, uint32_t symbol) {
   const uint8_t kind = s.kind[symbol];
   if (kind == PLANE_EXTRA) {
      return s.extra + (static_cast<size_t>(s.index[symbol]) * s.positions + position) * s.row_words;
   }
   return nullptr;
}

/// Where a position sits in the adaptive planes and how its codes read.
struct PositionLayout {
   const uint64_t* rows;  // first plane row
   uint32_t bits;         // plane rows: code planes, or one-hot rows
   bool identity;
   bool one_hot;
   bool implicit;         // one-hot rows with the symbol map[IMPLICIT_SLOT] derived
   const uint8_t* map;    // code (or 1 + one-hot row) -> scan symbol index (unused when identity)
};
__device__ __forceinline__ PositionLayout layoutOf(const SeqStoreDev& s, uint32_t position) {
   if (s.code_map == nullptr) {
      return {s.planes + static_cast<size_t>(position) * s.n_bits * s.row_words, s.n_bits, true, false, false, nullptr};
   }
   const uint8_t* map = s.code_map + static_cast<size_t>(position) * CODE_MAP_STRIDE;
   return {s.planes + static_cast<size_t>(s.row_of[position]) * s.row_words, static_cast<uint32_t>(map[0] & LAYOUT_ROWS_MASK), (map[0] & LAYOUT_IDENTITY) != 0,
           (map[0] & LAYOUT_ONE_HOT) != 0, (map[0] & LAYOUT_IMPLICIT) != 0, map};
}

/// The code (0 = none) that stands for scan symbol index `scan_index` at a position — for a one-hot position 1 + the row
/// that holds the symbol — or CODE_ESCAPED when the symbol has neither there (its rows are escape keys), CODE_IMPLICIT when
/// it is the position's derived symbol (no row, no keys).
constexpr uint32_t CODE_ESCAPED = 0xFFFFFFFFu;
constexpr uint32_t CODE_IMPLICIT = 0xFFFFFFFEu;
__device__ __forceinline__ uint32_t codeOfSymbol(const PositionLayout& layout, uint32_t scan_index) {
   if (layout.identity) {
      return scan_index + 1u;
   }
   if (layout.implicit && layout.map[IMPLICIT_SLOT] == scan_index) {
      return CODE_IMPLICIT;
   }
   const uint32_t n_codes = layout.one_hot ? layout.bits + 1u : (1u << layout.bits);
   for (uint32_t code = 1; code < n_codes; ++code) {
      if (layout.map[code] == scan_index) {
         return code;
      }
   }
   return CODE_ESCAPED;
}

/// Word `word` of the rows whose code at the position is `code`, decoded from the position's planes (one-hot: read).
__device__ __forceinline__ uint64_t decodeCodeWord(const PositionLayout& layout, uint32_t row_words, uint32_t code, uint32_t word) {
   if (layout.one_hot) {
      return layout.rows[static_cast<size_t>(code - 1u) * row_words + word];
   }
   uint64_t match = ~0ull;
   for (uint32_t bit = 0; bit < layout.bits; ++bit) {
      const uint64_t plane_word = layout.rows[static_cast<size_t>(bit) * row_words + word];
      match &= ((code >> bit) & 1u) != 0 ? plane_word : ~plane_word;
   }
   return match;  // padding bits have code 0, every coded symbol a code >= 1
}

/// The row's code at the position (0 = none coded): read out of the code planes, or the one-hot row that has its bit.
__device__ __forceinline__ uint32_t codeOfRow(const PositionLayout& layout, uint32_t row_words, uint32_t word, uint32_t bit) {
   uint32_t code = 0;
   for (uint32_t plane = 0; plane < layout.bits; ++plane) {
      const uint32_t set = static_cast<uint32_t>((layout.rows[static_cast<size_t>(plane) * row_words + word] >> bit) & 1u);
      code = layout.one_hot ? (set != 0 ? plane + 1u : code) : (code | (set << plane));
   }
   return code;
}

struct SeqStoreHost {
   SeqStoreDev dev{};
   uint32_t alphabet = 0;
   std::vector<uint8_t> reference;
   uint8_t* d_reference = nullptr;
   // sparse symbols: key = position << 37 | symbol << 32 | sequence id
   uint64_t* d_sparse = nullptr;
   uint32_t sparse_capacity = 0;
   uint32_t* d_sparse_count = nullptr;  // device counter
   std::vector<uint64_t> sparse_sorted;  // host copy after finalize
   bool finalized = false;
   // rows that received a sequence (append / generate; an import counts none: its bitmaps may leave rows without a symbol).  Only
   // a store whose every row has a symbol at every position may derive a symbol as "the rest" (LAYOUT_IMPLICIT).
   uint64_t rows_filled = 0;
   // counts of the unfiltered store, [positions][n_scan]: what the reference reads from stored
   // cardinalities for a full filter (mutations.cpp:98-136); computed by one scan on first use
   uint32_t* d_totals = nullptr;
   bool totals_ready = false;
   // the runs of the missing symbol (PLANE_RUNS), owned
   uint64_t* d_missing_run_keys = nullptr;
   uint32_t* d_missing_run_ends = nullptr;
   // a two-pass build between its passes / during the second (silo_gpu_store_build_pass): the layout in the making
   struct LayoutWork;
   std::shared_ptr<LayoutWork> work;
   unsigned long long* d_run_count = nullptr;  // runs of the missing symbol counted / written while the store is built in two passes
   // The adaptive code planes of the finalized store (see SeqStoreDev and buildLayout).
   struct Run {  // consecutive positions of one layout: a scan launch takes runs of ONE layout
      uint32_t begin;
      uint32_t end;
      uint8_t bits;   // code planes per position; 0 for a run of one-hot positions (1..3 rows each)
      bool identity;
      bool one_hot;
   };
   struct Layout {
      bool built = false;
      uint64_t* planes = nullptr;       // owned; nullptr when the store keeps its build-time planes (dev.planes == dev.scan)
      uint32_t* d_row_of = nullptr;
      uint32_t* d_row_target = nullptr;  // [rows] one-hot rows: position * n_scan + scan symbol of the row (else 0xFFFFFFFF)
      uint8_t* d_code_map = nullptr;
      uint64_t* d_escapes = nullptr;
      uint32_t* d_escape_first = nullptr;
      // the same keys once more, SLICE-major: slice = sequence >> slice_shift, (position, symbol, sequence) order within a
      // slice — what the scan's escape pass streams, a slice of the filter in LDS (k_scan_escapes_sliced)
      // PACKED, 4 bytes per key: row within the slice (17 bits) | counter relative to the key's granule << 17, where a granule is
      // ESCAPE_GRANULE_KEYS consecutive keys of one slice (a slice's keys are padded to whole granules with 0xFFFFFFFF) and
      // granule_base[g] = the counter (position * n_scan + symbol) of the granule's first key.  A key whose relative counter does
      // not fit 15 bits (a stretch of positions almost without keys) goes, 8 bytes wide, to the overflow list instead.
      uint32_t* d_escapes_sliced = nullptr;
      uint32_t* d_granule_base = nullptr;          // [packed keys / ESCAPE_GRANULE_KEYS]
      uint64_t* d_escapes_overflow = nullptr;      // counter << 32 | sequence
      uint32_t n_overflow = 0;
      uint64_t packed_keys = 0;                    // slots of d_escapes_sliced (keys + padding)
      uint32_t slice_shift = 0;
      uint32_t n_slices = 0;
      uint32_t* d_slice_first = nullptr;          // [n_slices][P + 1] first key of a position within a slice
      std::vector<uint32_t> slice_first;          // host copy
      std::vector<uint32_t> row_of;               // [P + 1]
      std::vector<uint8_t> code_map;              // [P][CODE_MAP_STRIDE]
      std::vector<uint32_t> escape_first;         // [P + 1]
      std::vector<uint32_t> escape_first_symbol;  // [P * n_scan + 1]: first key of a (position, scan symbol)
      std::vector<Run> runs;
      uint64_t device_bytes = 0;
      // positions whose most numerous symbol is derived (LAYOUT_IMPLICIT): a scan then counts the rows of the filter without a
      // valid symbol per position — the runs of the missing symbol by slices of 2^17 sequences (run_slice_first), the sparse keys
      bool has_implicit = false;
      uint32_t* d_run_slice_first = nullptr;  // [n_run_slices + 1] first run of a slice of sequences
      uint32_t n_run_slices = 0;
   } layout;
};

/// A layout in the making (between planLayout and finishLayout): the host tables, the device arrays the finished store will
/// own, and the two the encoders need on top (first key and cursor of every (position, symbol)).
struct SeqStoreHost::LayoutWork {
   std::vector<uint8_t> code_map;
   std::vector<uint32_t> row_of, row_target, escape_first, escape_first_symbol;
   std::vector<Run> runs;
   uint64_t total_rows = 0, total_escapes = 0;
   size_t plane_bytes = 0, escape_bytes = 0;
   bool has_implicit = false;
   uint8_t* d_code_map = nullptr;
   uint32_t* d_cursor = nullptr;
   uint32_t* d_first = nullptr;
   uint32_t* d_row_of = nullptr;
   uint32_t* d_row_target = nullptr;
   uint32_t* d_escape_first = nullptr;
   uint64_t* d_planes = nullptr;
   uint64_t* d_escapes = nullptr;
   void discard() {
      (void)hipFree(d_row_target);
      (void)hipFree(d_code_map);
      (void)hipFree(d_cursor);
      (void)hipFree(d_first);
      (void)hipFree(d_row_of);
      (void)hipFree(d_escape_first);
      (void)hipFree(d_planes);
      (void)hipFree(d_escapes);
      *this = LayoutWork{};
   }
};


}  // namespace silo_gpu_detail

struct silo_gpu_store {
   int device = 0;
   uint32_t sequence_count = 0;
   uint32_t row_words = 0;
   uint64_t device_bytes = 0;
   std::vector<silo_gpu_detail::SeqStoreHost> seqstores;
   uint64_t* d_ones = nullptr;       // the Full bitset
   uint16_t* d_lineage = nullptr;    // synthetic stores only
   uint32_t n_lineages = 0;
   uint32_t* d_error_flag = nullptr;
   // staging of append_sequences, grown on demand and reused across batches
   uint8_t* d_stage = nullptr;
   size_t stage_capacity = 0;
   uint8_t* d_stage_null = nullptr;
   size_t stage_null_capacity = 0;
   // scratch of silo_gpu_store_import_position: the one-hot row being expanded and the union of the rows seen so far
   uint64_t* d_import_row = nullptr;
   uint64_t* d_import_union = nullptr;
   // how finalize lays the store out: this store's choice, or (SILO_GPU_OPTION_DEFAULT) the process-wide silo_gpu_tune knob
   silo_gpu_store_options options{SILO_GPU_OPTION_DEFAULT, SILO_GPU_OPTION_DEFAULT, SILO_GPU_OPTION_DEFAULT, SILO_GPU_OPTION_DEFAULT};
   uint8_t* d_char_table[2] = {nullptr, nullptr};  // per alphabet, uploaded on first use
   char* d_symbol_chars[2] = {nullptr, nullptr};   // symbol -> character, for FastaAligned
   std::mutex mutex;
};

namespace silo_gpu_detail {

/// The layout options of a store: its own, or the process-wide knob where it has none.
inline int layoutOption(const silo_gpu_store* store) {
   return store->options.layout != SILO_GPU_OPTION_DEFAULT ? store->options.layout : g_tune_compact_index.load();
}
inline int missingRunsOption(const silo_gpu_store* store) {
   return store->options.missing_runs != SILO_GPU_OPTION_DEFAULT ? store->options.missing_runs : g_tune_missing_runs.load();
}
inline int keyCostOption(const silo_gpu_store* store) {
   return store->options.key_cost != SILO_GPU_OPTION_DEFAULT ? store->options.key_cost : g_tune_key_cost.load();
}
inline int launchCostOption(const silo_gpu_store* store) {
   return store->options.launch_cost_kib != SILO_GPU_OPTION_DEFAULT ? store->options.launch_cost_kib : g_tune_launch_cost.load();
}

// ------------------------------------------------------------------------------------------------
// wave-level helpers
// ------------------------------------------------------------------------------------------------
// Inclusive DPP scan within rows of 16, then row broadcasts: lane 63 ends up with the wave sum.
// 6 VALU instructions, no LDS traffic (ds_bpermute-based __shfl costs an LDS round trip per step).
__device__ __forceinline__ uint32_t waveSumToLane63(uint32_t v) {
   v += __builtin_amdgcn_update_dpp(0u, v, 0x111, 0xf, 0xf, false);  // row_shr:1
   v += __builtin_amdgcn_update_dpp(0u, v, 0x112, 0xf, 0xf, false);  // row_shr:2
   v += __builtin_amdgcn_update_dpp(0u, v, 0x114, 0xf, 0xf, false);  // row_shr:4
   v += __builtin_amdgcn_update_dpp(0u, v, 0x118, 0xf, 0xf, false);  // row_shr:8
   v += __builtin_amdgcn_update_dpp(0u, v, 0x142, 0xa, 0xf, false);  // row_bcast:15 -> rows 1,3
   v += __builtin_amdgcn_update_dpp(0u, v, 0x143, 0xc, 0xf, false);  // row_bcast:31 -> rows 2,3
   return v;
}


constexpr int SCAN_THREADS = 256;  // threads of a k_scan_sliced block; a store is re-encoded where a row fills at least one of its column tiles

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

// 16-byte plane load; NT marks the stream non-temporal (read once, keep it out of the way of the filter tile).
template <bool NT>
__device__ __forceinline__ ulonglong2 loadPlane16(const uint64_t* ptr) {
   if constexpr (NT) {
      const u32x4 v = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(ptr));
      return make_ulonglong2(
         static_cast<uint64_t>(v.x) | (static_cast<uint64_t>(v.y) << 32), static_cast<uint64_t>(v.z) | (static_cast<uint64_t>(v.w) << 32)
      );
   } else {
      return *reinterpret_cast<const ulonglong2*>(ptr);
   }
}

// 16-byte load through a pointer that is known to point into device memory but was itself read from memory (a table
// of leaf pointers): without the explicit global address space the compiler has to emit flat_load, which may alias LDS —
// every such load is then fenced against the slot accesses around it (s_waitcnt vmcnt(0) lgkmcnt(0)) and runs of
// independent leaf loads are serialised.
typedef unsigned long long u64x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ ulonglong2 loadGlobal16(const uint64_t* ptr) {
   const u64x2 v = *(const __attribute__((address_space(1))) u64x2*)(ptr);
   return make_ulonglong2(v.x, v.y);
}

/// The escape pass over the SLICE-major copy of the keys.  A key costs one filter-bit lookup, and 64 lanes looking up 64
/// rows of a 1.25 MB filter pull 64 cache lines through the L2 for 64 bits (45 M keys: 5.8 GB of line traffic, 0.28 ms —
/// as much as 40 plane bytes per key).  Here a block owns one slice of the rows, copies that slice of the filter into LDS
/// (16 KiB for 2^17 rows) and streams the slice's keys of the scanned positions against it: a lookup is an LDS read.
constexpr uint32_t ESCAPE_SLICE_SHIFT = 17;                    // 2^17 rows = 2048 filter words = 16 KiB of LDS per filter
constexpr uint32_t ESCAPE_SLICE_WORDS32 = (1u << ESCAPE_SLICE_SHIFT) / 32u;
constexpr uint32_t ESCAPE_SLICE_BITS = 9;                      // sequence bits that number the slices
constexpr uint32_t ESCAPE_MAX_SLICES = 1u << ESCAPE_SLICE_BITS;  // 67 M rows
constexpr uint32_t ESCAPE_SLICE_THREADS = 1024;
constexpr uint32_t ESCAPE_MAX_RANGES = 16;
constexpr uint32_t ESCAPE_GRANULE_KEYS = 4096;                 // keys that share a base counter: one 16-byte load per lane of a 1024-thread block
constexpr uint32_t ESCAPE_ROW_MASK = (1u << ESCAPE_SLICE_SHIFT) - 1u;
constexpr uint32_t ESCAPE_KEY_INVALID = 0xFFFFFFFFu;           // padding / a key that went to the overflow list
// largest relative counter a packed key holds (15 bits would allow 32 766): small enough that a granule by itself always fits the
// window of LDS counters of k_scan_escapes_sliced<1> and <2> (6 144 counters less a position's worth), so that those count without
// a path for keys past the window; the few keys further behind their granule's first go to the overflow list
constexpr uint32_t ESCAPE_MAX_RELATIVE = 6000u;

/// A position range of one sequence store with the count tables of every filter of the launch.
struct ScanRange {
   const SeqStoreHost* seqstore;
   uint32_t pos_begin;
   uint32_t pos_end;
   uint32_t* counts[SILO_GPU_MAX_SCAN_BATCH];
};

/// The Mutations scan of `q_count` filters over position ranges of sequence stores of one alphabet (silo_gpu_scan.hip); finalize
/// uses it for the unfiltered totals that decide the layout.
int scanRanges(const silo_gpu_store* store, const std::vector<ScanRange>& ranges, const uint64_t* const* filters, uint32_t q_count, hipStream_t hip_stream);

// host helpers of the store (silo_gpu_store.hip)
int ensureDevice(int device);
int growSparse(SeqStoreHost& seqstore, uint32_t needed);
int ensureBuildPlanes(silo_gpu_store* store, SeqStoreHost& seqstore);

}  // namespace silo_gpu_detail
