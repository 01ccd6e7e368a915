// silo_gpu.hip — CDNA4 (gfx950) kernels + C ABI of the SILO mutation-filter hot path.
//
// Kernels:
//   K1  k_scan_tiled / k_scan_rowwave   Mutations scan: counts[p][s] += popcount(F & C[p][s])
//                                        (reference: actions/mutations.cpp:64-164)
//   K2  k_popcount                      |F|            (actions/aggregated.cpp:61)
//   K3  k_filter_eval                   fused operator tree -> bitset (+ count)
//                                        (operators/{index_scan,complement,intersection,union,
//                                         threshold,full,empty,bitmap_selection}.cpp)
//   B1  k_transpose_sequences           aligned sequences -> bit planes (storage/sequence_store.cpp:100-190)
//   B2  k_generate_synthetic            synthetic planes for the benchmarks
// Everything is 64-bit integer AND / OR / popcount: HBM-bound, no MFMA.
#include <hip/hip_runtime.h>

#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include <algorithm>
#include <array>
#include <atomic>
#include <memory>
#include <mutex>
#include <new>
#include <string>
#include <vector>

#include "../../include/silo_gpu.h"
#include "bitprog.h"
#include "internal.h"
#include "layout_choice.h"

namespace {

thread_local std::string g_last_error;
thread_local const char* g_last_scan_kernel = "none";

std::atomic<int> g_tune_rows_per_block{0};
std::atomic<int> g_tune_scan_variant{0};
std::atomic<int> g_tune_eval_leaf_batch{0};
std::atomic<int> g_tune_compact_index{0};  // < 0: never scan the compact index (K1i), even where one was built
std::atomic<int> g_tune_side_stream{0};     // the escape pass: 0 = side stream of the lowest priority, 1 = of default priority, 2 = the caller's stream, 3 = as 0 with the position-major keys (k_scan_escapes)
std::atomic<int> g_tune_scan_timing{0};     // 1: HIP events around every plane-scan launch (silo_gpu_scan_timings)
std::atomic<int> g_tune_missing_runs{0};    // < 0: finalize keeps the plane of the missing symbol instead of turning it into runs
std::atomic<int> g_tune_key_cost{0};        // > 0: what an escape key costs in plane bytes in the layout choice (default KEY_COST_BYTES)
std::atomic<int> g_tune_launch_cost{0};     // what a further kind of plane-scan launch costs in the layout choice, in KiB of plane bytes: 0 = default (LAUNCH_COST_BYTES), < 0 = nothing
std::atomic<int> g_tune_sparse_divisor{0};  // 0 = default (row_words / 16 filter sectors with a set bit), < 0 = sparse-filter path off

int fail(int code, const std::string& msg) {
   g_last_error = msg;
   return code;
}

}  // namespace

int silo_gpu_internal_fail(int code, const std::string& message) {  // for the other translation units (internal.h)
   return fail(code, message);
}

namespace {

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
void fillCharTable(uint32_t alphabet, uint8_t table[256]) {
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

uint32_t alphabetSize(uint32_t alphabet) {
   return alphabet == SILO_GPU_ALPHABET_NUCLEOTIDE ? SILO_GPU_NUC_SYMBOLS : SILO_GPU_AA_SYMBOLS;
}
uint32_t missingSymbol(uint32_t alphabet) {
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
      uint64_t* d_escapes_sliced = nullptr;
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

}  // namespace

struct silo_gpu_store {
   int device = 0;
   uint32_t sequence_count = 0;
   uint32_t row_words = 0;
   uint64_t device_bytes = 0;
   std::vector<SeqStoreHost> seqstores;
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

namespace {

/// The layout options of a store: its own, or the process-wide knob where it has none.
int layoutOption(const silo_gpu_store* store) {
   return store->options.layout != SILO_GPU_OPTION_DEFAULT ? store->options.layout : g_tune_compact_index.load();
}
int missingRunsOption(const silo_gpu_store* store) {
   return store->options.missing_runs != SILO_GPU_OPTION_DEFAULT ? store->options.missing_runs : g_tune_missing_runs.load();
}
int keyCostOption(const silo_gpu_store* store) {
   return store->options.key_cost != SILO_GPU_OPTION_DEFAULT ? store->options.key_cost : g_tune_key_cost.load();
}
int launchCostOption(const silo_gpu_store* store) {
   return store->options.launch_cost_kib != SILO_GPU_OPTION_DEFAULT ? store->options.launch_cost_kib : g_tune_launch_cost.load();
}

int buildLayout(silo_gpu_store* store, SeqStoreHost& seqstore);  // the adaptive code planes, defined next to the scan launchers
bool reencodes(const silo_gpu_store* store, const SeqStoreDev& dev);
int planLayout(const silo_gpu_store* store, SeqStoreHost& seqstore, SeqStoreHost::LayoutWork& work, bool zero_planes, bool allow_implicit, bool* fits);

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

// ------------------------------------------------------------------------------------------------
// K1: Mutations scan over the bit-sliced planes.
//
// counts[q][p][k] += popcount(filter_q & {rows whose code at position p is k + 1}) for the NSYM valid mutation symbols,
// reading BITS = ceil(log2(NSYM + 1)) planes per position (3 for nucleotides, 5 for amino acids) instead of NSYM
// one-hot planes: 0.375 instead of 0.625 bytes per position x sequence (nuc), 0.625 instead of 2.75 (aa).
//
// Grid: blockIdx.x = position_group * n_tiles + tile.  A block owns a column tile of TILE_WORDS = 256 threads * WPT
// words of the Q filters, held in registers for the whole block lifetime (registers are the first-level staging of
// the filter, LDS only carries per-wave partial counts), and streams that tile's slice of the BITS plane rows of
// `positions_per_block` consecutive positions.  Every load is a fully coalesced, non-temporal 16 B/lane access; the
// planes of position p+1 are in flight while position p is decoded (two register buffers, unconditional clamped
// loads so that s_waitcnt keeps counting).  Decoding is pure VALU: per symbol BITS and/andn per word (constant-folded
// code bits, shared sub-terms), an AND with each filter, v_bcnt; then a 6-instruction DPP wave reduction per
// (symbol, filter).  Out-of-row chunks of the ragged last tile read word 0 against zero filters.
// ------------------------------------------------------------------------------------------------
constexpr int SCAN_THREADS = 256;
constexpr int SCAN_WAVES = SCAN_THREADS / 64;

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

constexpr uint32_t SCAN_MAX_RANGES = 32;
// the per-filter sector counters sit 256 bytes apart: atomics on one L2 channel serialise (~12 ns each), and a dense
// filter makes every block add to its counter
constexpr uint32_t SPARSE_COUNTER_STRIDE = 64;
constexpr uint32_t SECTOR_WORDS = 8;        // a 64-byte sector of a filter row
constexpr uint32_t COMPACT_THREADS = 1024;   // words per block of k_compact_filter

/// Which scan serves a filter, from the counters k_compact_filter left for it: [0] sectors with a set bit, [1] stretches of
/// COMPACT_THREADS words with one.  The gather pays while the sectors fit its list AND cost less than the column tiles the
/// dense scan cannot skip: it reads its sectors at about 0.6 of the dense scan's rate, so a clustered filter (rows in
/// lineage or date order: few sectors because they are contiguous, not because they are few) stays with the dense scan.
__device__ __forceinline__ bool takesGatherScan(const uint32_t* __restrict__ counters, uint32_t capacity) {
   const uint32_t sectors = counters[0];
   return sectors <= capacity && static_cast<uint64_t>(sectors) * 8u < static_cast<uint64_t>(counters[1]) * (COMPACT_THREADS / SECTOR_WORDS) * 5u;
}


/// One launch of the scan: up to SILO_GPU_MAX_SCAN_BATCH filters against up to SCAN_MAX_RANGES position ranges of
/// sequence stores with the same layout (the 12 genes of an AminoAcidMutations query, the segments of a segmented
/// genome): blocks (k_scan_sliced) or waves (k_scan_gather) are dealt to the ranges by first_unit.
struct ScanBatchArgs {
   const uint64_t* filters[SILO_GPU_MAX_SCAN_BATCH];
   // sparse-filter routing (K1s): sparse_sectors[q * SPARSE_COUNTER_STRIDE] = number of 64-byte sectors of filter q with a set bit, written by
   // k_compact_filter earlier on the same stream; a filter with at most sparse_capacity of them is served by
   // k_scan_gather and is treated as empty by k_scan_sliced.  nullptr = no routing.
   const uint32_t* sparse_sectors;
   uint32_t sparse_capacity;
   uint32_t n_ranges;
   const uint64_t* planes[SCAN_MAX_RANGES];    // first plane row of the range
   uint32_t n_positions[SCAN_MAX_RANGES];
   uint32_t first_unit[SCAN_MAX_RANGES + 1];   // prefix sums of the blocks / waves per range
   uint32_t* counts[SCAN_MAX_RANGES][SILO_GPU_MAX_SCAN_BATCH];  // counts[range][filter], at the first position of the range
   // mapped layouts (2 or 3 code planes): per position of the range CODE_MAP_STRIDE bytes, [c] = the scan symbol that
   // code c stands for at this position (0xFF = none); out_symbols = symbols per position of the count tables (5 / 22)
   // one-hot rows (KIND_ROWS): the range is a run of plane ROWS, n_positions counts rows, code_map[range] points at the
   // uint32 table row -> position * out_symbols + symbol (positions of the store), target_base = that of counts[range]
   const uint8_t* code_map[SCAN_MAX_RANGES];
   uint32_t target_base[SCAN_MAX_RANGES];
   uint32_t out_symbols;
};

// what a run of plane rows holds
enum : int { KIND_IDENTITY = 0, KIND_MAPPED = 1, KIND_ROWS = 2 };

// positions whose partial counts sit in LDS between two flushes: ~16 KiB of LDS whatever NSYM * Q is
template <int NSYM, int Q>
constexpr int scanPositionsBatch() {
   int batch = 512 / (NSYM * Q);
   batch -= batch & 1;
   return batch < 2 ? 2 : (batch > 64 ? 64 : batch);
}

// blocks per CU the register budget has to allow: plane buffers 2 * BITS * WPT * 2 VGPRs, filters Q * WPT * 2
template <int BITS, int NSYM, int WPT, int Q>
constexpr int scanMinBlocks() {
   if (BITS == 3 && NSYM == 7 && WPT == 8) {
      return 2;  // 7 counted symbols over 8 words per thread: 3 blocks per CU would spill
   }
   return Q == 1 ? (BITS * WPT <= 12 ? 4 : (BITS * WPT <= 18 ? 4 : (BITS <= 3 && BITS * WPT <= 24 ? 3 : 2))) : (Q <= 2 && BITS <= 3 ? 4 : (Q <= 4 && BITS <= 3 ? 3 : 2));
}

template <int BITS, int NSYM, int WPT, int Q, int KIND>
__global__ __launch_bounds__(SCAN_THREADS, (scanMinBlocks<BITS, NSYM, WPT, Q>())) void k_scan_sliced(
   const ScanBatchArgs batch, uint32_t row_words, uint32_t positions_per_block, uint32_t n_tiles
) {
   constexpr int CHUNKS = WPT / 2;  // 16-byte chunks per thread and plane
   constexpr uint32_t TILE_WORDS = SCAN_THREADS * WPT;
   constexpr int POS_BATCH = scanPositionsBatch<NSYM, Q>();
   __shared__ uint32_t s_partial[2][SCAN_WAVES][POS_BATCH][NSYM * Q];

   const uint32_t tid = threadIdx.x;
   const uint32_t wave = tid >> 6;
   const bool writer = (tid & 63u) == 63u;  // waveSumToLane63 leaves the total in lane 63
   uint32_t range = 0;
   while (range + 1 < batch.n_ranges && blockIdx.x >= batch.first_unit[range + 1]) {
      ++range;
   }
   const uint32_t block_in_range = blockIdx.x - batch.first_unit[range];
   const uint64_t* __restrict__ planes = batch.planes[range];
   // one-hot rows: a "position" of the pipeline is a PAIR of rows (BITS = NSYM = 2), each counted on its own
   static_assert(KIND != KIND_ROWS || (BITS == 2 && NSYM == 2), "rows are scanned in pairs");
   const uint32_t n_rows = batch.n_positions[range];
   const uint32_t n_positions = KIND == KIND_ROWS ? (n_rows + 1u) / 2u : n_rows;
   const uint32_t tile = block_in_range % n_tiles;
   const uint32_t position_group = block_in_range / n_tiles;
   const uint32_t pos_begin = position_group * positions_per_block;
   const uint32_t pos_end = min(n_positions, pos_begin + positions_per_block);
   const uint32_t last_pos = pos_end - 1;

   // filters routed to the gather kernel count as empty here; a block with nothing left to do leaves at once
   bool dense[Q];
#pragma unroll
   for (int q = 0; q < Q; ++q) {
      dense[q] = batch.sparse_sectors == nullptr || !takesGatherScan(batch.sparse_sectors + q * SPARSE_COUNTER_STRIDE, batch.sparse_capacity);
   }
   bool any_dense = false;
#pragma unroll
   for (int q = 0; q < Q; ++q) {
      any_dense |= dense[q];
   }
   if (!any_dense) {
      return;
   }

   // this thread's 16-byte chunks of the tile; the filter words stay in registers for all positions
   uint32_t word[CHUNKS];
   ulonglong2 f[Q][CHUNKS];
#pragma unroll
   for (int j = 0; j < CHUNKS; ++j) {
      word[j] = tile * TILE_WORDS + (j * SCAN_THREADS + tid) * 2;
      const bool inside = word[j] < row_words;
      if (!inside) {
         word[j] = 0;  // out-of-row chunks read word 0 (always valid) against zero filters: no branch in the loop
      }
#pragma unroll
      for (int q = 0; q < Q; ++q) {
         f[q][j] = inside && dense[q] ? *reinterpret_cast<const ulonglong2*>(batch.filters[q] + word[j]) : make_ulonglong2(0, 0);
      }
   }

   // A tile without a selected row has nothing to count: rows laid out by lineage or date (the reference partitions by a
   // key column and orders by date, preprocessor.cpp:159-227) give lineage and date filters long runs of zero words, and such a block leaves before its first load.
   {
      uint64_t any_bit = 0;
#pragma unroll
      for (int j = 0; j < CHUNKS; ++j) {
#pragma unroll
         for (int q = 0; q < Q; ++q) {
            any_bit |= f[q][j].x | f[q][j].y;
         }
      }
      if (__syncthreads_or(any_bit != 0 ? 1 : 0) == 0) {
         return;
      }
   }

   auto load_position = [&](uint32_t position, ulonglong2 (&dst)[BITS][CHUNKS]) {
      const uint64_t* base = planes + static_cast<size_t>(position) * BITS * row_words;
#pragma unroll
      for (int bit = 0; bit < BITS; ++bit) {
         // the second row of the last pair of an odd run is the first one again (in bounds, not stored)
         const size_t row = KIND == KIND_ROWS ? static_cast<size_t>(min(static_cast<uint32_t>(bit), n_rows - 1u - position * 2u)) : static_cast<size_t>(bit);
#pragma unroll
         for (int j = 0; j < CHUNKS; ++j) {
            dst[bit][j] = loadPlane16<true>(base + row * row_words + word[j]);
         }
      }
   };
   auto reduce_position = [&](const ulonglong2 (&src)[BITS][CHUNKS], uint32_t buffer, uint32_t slot, bool store) {
      uint32_t acc[NSYM][Q];
#pragma unroll
      for (int symbol = 0; symbol < NSYM; ++symbol) {
#pragma unroll
         for (int q = 0; q < Q; ++q) {
            acc[symbol][q] = 0;
         }
      }
#pragma unroll
      for (int j = 0; j < CHUNKS; ++j) {
#pragma unroll
         for (int half = 0; half < 2; ++half) {
            uint64_t bits[BITS];
#pragma unroll
            for (int bit = 0; bit < BITS; ++bit) {
               bits[bit] = half == 0 ? src[bit][j].x : src[bit][j].y;
            }
            // Decode tree: the four combinations of the two low code bits, of the next two, and the top bit — a symbol
            // is then two ANDs (22 symbols from 5 planes: ~55 logic ops per word instead of 110).  With one filter the
            // filter is folded into the low pair, so the per-symbol AND with it disappears as well.
            const uint64_t filter0 = half == 0 ? f[0][j].x : f[0][j].y;
            if constexpr (KIND == KIND_ROWS) {
#pragma unroll
               for (int row = 0; row < NSYM; ++row) {
#pragma unroll
                  for (int q = 0; q < Q; ++q) {
                     acc[row][q] += static_cast<uint32_t>(__popcll(bits[row] & (half == 0 ? f[q][j].x : f[q][j].y)));
                  }
               }
               continue;
            }
            uint64_t low[4];
            low[0] = ~bits[1] & ~bits[0];
            low[1] = ~bits[1] & bits[0];
            low[2] = bits[1] & ~bits[0];
            low[3] = bits[1] & bits[0];
            if constexpr (Q == 1) {
#pragma unroll
               for (int k = 0; k < 4; ++k) {
                  low[k] &= filter0;
               }
            }
            uint64_t high[BITS <= 3 ? 2 : 8];
            static_assert(NSYM < (1 << BITS), "every counted code needs a bit pattern of its own, 0 is 'none'");
            if constexpr (BITS == 2) {
               high[0] = ~0ull;  // the codes ARE the low pair
               high[1] = 0;
            } else if constexpr (BITS == 3) {
               high[0] = ~bits[2];
               high[1] = bits[2];
            } else {
               static_assert(BITS == 5, "decode tree written for 2, 3 or 5 code bits");
#pragma unroll
               for (int k = 0; k < 8; ++k) {
                  high[k] = ((k & 1) != 0 ? bits[2] : ~bits[2]) & ((k & 2) != 0 ? bits[3] : ~bits[3]) & ((k & 4) != 0 ? bits[4] : ~bits[4]);
               }
            }
#pragma unroll
            for (int symbol = 0; symbol < NSYM; ++symbol) {
               const uint32_t code = static_cast<uint32_t>(symbol) + 1u;
               const uint64_t match = BITS == 2 ? low[code & 3u] : (low[code & 3u] & high[code >> 2]);
#pragma unroll
               for (int q = 0; q < Q; ++q) {
                  const uint64_t filter_word = half == 0 ? f[q][j].x : f[q][j].y;
                  acc[symbol][q] += static_cast<uint32_t>(__popcll(Q == 1 ? match : (match & filter_word)));
               }
            }
         }
      }
      // wave reduction, two symbols per register: a lane counted at most WPT * 64 <= 512 rows per symbol, so a wave total
      // fits 16 bits (<= 32 768) and the 6 DPP steps serve two symbols at once
      static_assert(WPT * 64 * 64 < 65536, "packed wave totals need 16 bits per symbol");
#pragma unroll
      for (int symbol = 0; symbol < NSYM; symbol += 2) {
#pragma unroll
         for (int q = 0; q < Q; ++q) {
            const bool pair = symbol + 1 < NSYM;
            const uint32_t packed = pair ? (acc[symbol][q] | (acc[symbol + 1 < NSYM ? symbol + 1 : symbol][q] << 16)) : acc[symbol][q];
            const uint32_t total = waveSumToLane63(packed);
            if (writer && store) {
               s_partial[buffer][wave][slot][q * NSYM + symbol] = pair ? (total & 0xFFFFu) : total;
               if (pair) {
                  s_partial[buffer][wave][slot][q * NSYM + symbol + 1] = total >> 16;
               }
            }
         }
      }
   };
   auto flush = [&](uint32_t batch_first_position, uint32_t n_batch, uint32_t buffer) {
      __syncthreads();
      for (uint32_t item = tid; item < n_batch * (NSYM * Q); item += SCAN_THREADS) {
         const uint32_t position = item / (NSYM * Q);
         const uint32_t rest = item % (NSYM * Q);
         uint32_t total = 0;
#pragma unroll
         for (int w = 0; w < SCAN_WAVES; ++w) {
            total += s_partial[buffer][w][position][rest];
         }
         if (total != 0) {
            if constexpr (KIND == KIND_ROWS) {  // row -> its (position, symbol) counter
               const uint32_t row = (batch_first_position + position) * 2u + rest % NSYM;
               if (row < n_rows) {
                  const uint32_t target = reinterpret_cast<const uint32_t*>(batch.code_map[range])[row] - batch.target_base[range];
                  atomicAdd(&batch.counts[range][rest / NSYM][target], total);
               }
            } else if constexpr (KIND == KIND_MAPPED) {  // code -> the symbol it stands for at this position
               const uint32_t symbol = batch.code_map[range][static_cast<size_t>(batch_first_position + position) * CODE_MAP_STRIDE + 1 + rest % NSYM];
               if (symbol < batch.out_symbols) {  // an unused code (0xFF) has no rows: never taken, never out of bounds
                  atomicAdd(&batch.counts[range][rest / NSYM][static_cast<size_t>(batch_first_position + position) * batch.out_symbols + symbol], total);
               }
            } else {
               atomicAdd(&batch.counts[range][rest / NSYM][static_cast<size_t>(batch_first_position + position) * NSYM + rest % NSYM], total);
            }
         }
      }
   };

   ulonglong2 buf_a[BITS][CHUNKS];
   ulonglong2 buf_b[BITS][CHUNKS];
   load_position(pos_begin, buf_a);
   uint32_t buffer = 0;
   uint32_t batch_first_position = pos_begin;
   for (uint32_t position = pos_begin; position < pos_end; position += 2) {
      load_position(min(position + 1, last_pos), buf_b);
      reduce_position(buf_a, buffer, position - batch_first_position, true);
      load_position(min(position + 2, last_pos), buf_a);
      reduce_position(buf_b, buffer, position + 1 - batch_first_position, position + 1 < pos_end);
      const uint32_t done = min(position + 2, pos_end) - batch_first_position;
      if (done >= static_cast<uint32_t>(POS_BATCH) || position + 2 >= pos_end) {  // POS_BATCH is even
         flush(batch_first_position, done, buffer);
         batch_first_position += done;
         buffer ^= 1u;
      }
   }
}

// ------------------------------------------------------------------------------------------------
// K1b: one wave per position, for short rows (small N) where a 256-thread column tile would be mostly empty.
// ------------------------------------------------------------------------------------------------
template <int BITS, int NSYM>
__global__ __launch_bounds__(256) void k_scan_sliced_rowwave(
   const uint64_t* __restrict__ planes, const uint64_t* __restrict__ filter, uint32_t* __restrict__ counts, uint32_t row_words,
   uint32_t n_positions
) {
   const uint32_t lane = threadIdx.x & 63u;
   const uint32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
   const uint32_t n_waves = (gridDim.x * blockDim.x) >> 6;
   for (uint32_t position = wave; position < n_positions; position += n_waves) {
      const uint64_t* base = planes + static_cast<size_t>(position) * BITS * row_words;
      uint32_t acc[NSYM];
#pragma unroll
      for (int symbol = 0; symbol < NSYM; ++symbol) {
         acc[symbol] = 0;
      }
      for (uint32_t w = lane; w < row_words; w += 64) {
         const uint64_t filter_word = filter[w];
         uint64_t bits[BITS];
#pragma unroll
         for (int bit = 0; bit < BITS; ++bit) {
            bits[bit] = base[static_cast<size_t>(bit) * row_words + w];
         }
#pragma unroll
         for (int symbol = 0; symbol < NSYM; ++symbol) {
            const uint32_t code = static_cast<uint32_t>(symbol) + 1u;
            uint64_t match = filter_word;
#pragma unroll
            for (int bit = 0; bit < BITS; ++bit) {
               match &= ((code >> bit) & 1u) != 0 ? bits[bit] : ~bits[bit];
            }
            acc[symbol] += static_cast<uint32_t>(__popcll(match));
         }
      }
#pragma unroll
      for (int symbol = 0; symbol < NSYM; ++symbol) {
         const uint32_t total = waveSumToLane63(acc[symbol]);
         if (lane == 63u && total != 0) {
            atomicAdd(&counts[static_cast<size_t>(position) * NSYM + symbol], total);
         }
      }
   }
}

// ------------------------------------------------------------------------------------------------
// K1s: Mutations scan under a SPARSE filter.  The dense scan costs the same whatever the filter selects; the reference's
// roaring and_cardinality gets cheaper with the filter (mutations.cpp:139-164 over a small filter bitmap), so a query
// for a few hundred rows must not pay for 112 GB.  k_compact_filter lists the 64-byte SECTORS (8 consecutive words —
// the unit HBM delivers) of the filter that hold a set bit, at most `capacity` of them (the total is counted
// regardless); when they fit, k_scan_gather reads only those sectors of every plane and k_scan_sliced skips the
// filter.  The decision is taken on the device from the counters (takesGatherScan): no host round trip.  Measured at 10 M
// sequences (profiles/r01_sparse_filters.md, r02_one_hot_rows.md): ~0.9 µs per listed sector of the genome against 6 ms for
// the dense scan, hence the default capacity of row_words / 16 sectors.
// ------------------------------------------------------------------------------------------------

/// Also the scan's "prepare" step (one launch in front of everything else): the blocks zero `n_zero_words` words of scratch
/// (the private count tables of a scan with derived symbols) between them, add the filter's cardinality to counter [2], and
/// block (0, 0) zeroes the counter set the NEXT scan on this scratch block will use (the sets alternate: no fill launches).
__global__ __launch_bounds__(COMPACT_THREADS) void k_compact_filter(
   const ScanBatchArgs batch, uint32_t row_words, uint32_t capacity, uint32_t* __restrict__ sparse_sectors, uint32_t* __restrict__ sector_index,
   uint32_t* __restrict__ zero_words, uint32_t n_zero_words, uint32_t* __restrict__ counters_to_reset
) {
   __shared__ uint32_t s_wave_first[COMPACT_THREADS / 64];
   __shared__ uint32_t s_wave_rows[COMPACT_THREADS / 64];
   __shared__ uint32_t s_block_first;
   const uint32_t q = blockIdx.y;
   const uint32_t w = blockIdx.x * blockDim.x + threadIdx.x;  // row_words is a multiple of 32: sectors never straddle the row end
   const uint32_t lane = threadIdx.x & 63u;
   const uint32_t wave = threadIdx.x >> 6;
   const uint64_t value = w < row_words ? batch.filters[q][w] : 0;
   {  // this block's share of the scratch to zero (16-byte stores; n_zero_words is a multiple of 4)
      const uint32_t n_chunks = n_zero_words / 4u;
      const uint32_t n_threads = gridDim.x * gridDim.y * COMPACT_THREADS;
      for (uint32_t chunk = (blockIdx.y * gridDim.x + blockIdx.x) * COMPACT_THREADS + threadIdx.x; chunk < n_chunks; chunk += n_threads) {
         reinterpret_cast<uint4*>(zero_words)[chunk] = make_uint4(0, 0, 0, 0);
      }
      if (blockIdx.x == 0 && blockIdx.y == 0 && counters_to_reset != nullptr && threadIdx.x < SILO_GPU_MAX_SCAN_BATCH * SPARSE_COUNTER_STRIDE) {
         counters_to_reset[threadIdx.x] = 0;
      }
   }
   const uint64_t ballot = __ballot(value != 0);
   // one bit per sector of this wave (at the sector's first lane): does any of its 8 words have a set bit?
   uint64_t leaders = 0;
#pragma unroll
   for (uint32_t sector = 0; sector < 64 / SECTOR_WORDS; ++sector) {
      if (((ballot >> (sector * SECTOR_WORDS)) & 0xFFull) != 0) {
         leaders |= 1ull << (sector * SECTOR_WORDS);
      }
   }
   const uint32_t wave_rows = waveSumToLane63(static_cast<uint32_t>(__popcll(value)));
   if (lane == 0) {
      s_wave_first[wave] = static_cast<uint32_t>(__popcll(leaders));
   }
   if (lane == 63u) {
      s_wave_rows[wave] = wave_rows;
   }
   __syncthreads();
   if (threadIdx.x == 0) {  // exclusive prefix over the waves, ONE atomic per block
      uint32_t total = 0;
      uint32_t rows = 0;
      for (uint32_t k = 0; k < COMPACT_THREADS / 64; ++k) {
         const uint32_t count = s_wave_first[k];
         s_wave_first[k] = total;
         total += count;
         rows += s_wave_rows[k];
      }
      s_block_first = total != 0 ? atomicAdd(sparse_sectors + q * SPARSE_COUNTER_STRIDE, total) : 0;
      if (total != 0) {
         atomicAdd(sparse_sectors + q * SPARSE_COUNTER_STRIDE + 1, 1u);  // stretches of COMPACT_THREADS words with a set bit
         atomicAdd(sparse_sectors + q * SPARSE_COUNTER_STRIDE + 2, rows);  // the filter's cardinality
      }
   }
   __syncthreads();
   if (((leaders >> lane) & 1ull) != 0) {
      const uint32_t slot = s_block_first + s_wave_first[wave] + static_cast<uint32_t>(__popcll(leaders & ((1ull << lane) - 1ull)));
      if (slot < capacity) {
         sector_index[static_cast<size_t>(q) * capacity + slot] = w / SECTOR_WORDS;
      }
   }
}

// One WAVE per group of POSG consecutive positions (no LDS, no block-level reduction: a sparse filter may have fewer
// non-zero words than a block has lanes); lanes stride over the words of the listed sectors, POSG * BITS gathers in flight each.
template <int BITS, int NSYM, int POSG, int KIND>
__global__ __launch_bounds__(256, (BITS <= 3 ? (NSYM <= 5 ? 5 : 4) : 4)) void k_scan_gather(
   const ScanBatchArgs batch, const uint32_t* __restrict__ sector_index, uint32_t capacity, uint32_t row_words
) {
   const uint32_t q = blockIdx.y;
   const uint32_t n_sectors = batch.sparse_sectors[q * SPARSE_COUNTER_STRIDE];
   if (n_sectors == 0 || !takesGatherScan(batch.sparse_sectors + q * SPARSE_COUNTER_STRIDE, batch.sparse_capacity)) {
      return;  // empty filter, or a dense one (k_scan_sliced has it); `capacity` is the stride of the lists
   }
   const uint32_t n_words = n_sectors * SECTOR_WORDS;
   const uint32_t lane = threadIdx.x & 63u;
   const uint32_t unit = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;  // this wave
   if (unit >= batch.first_unit[batch.n_ranges]) {
      return;
   }
   uint32_t range = 0;
   while (range + 1 < batch.n_ranges && unit >= batch.first_unit[range + 1]) {
      ++range;
   }
   const uint64_t* __restrict__ planes = batch.planes[range];
   static_assert(KIND != KIND_ROWS || (BITS == 1 && NSYM == 1), "one-hot rows are gathered one by one");
   const uint32_t n_positions = batch.n_positions[range];  // KIND_ROWS: plane rows
   const uint32_t pos_begin = (unit - batch.first_unit[range]) * POSG;
   const uint32_t last_pos = n_positions - 1;
   const uint32_t* index = sector_index + static_cast<size_t>(q) * capacity;
   const uint64_t* filter = batch.filters[q];
   const size_t position_stride = static_cast<size_t>(BITS) * row_words;

   uint32_t acc[POSG][NSYM];
#pragma unroll
   for (int g = 0; g < POSG; ++g) {
#pragma unroll
      for (int symbol = 0; symbol < NSYM; ++symbol) {
         acc[g][symbol] = 0;
      }
   }
   for (uint32_t i = lane; i < n_words; i += 64) {
      const uint32_t w = index[i / SECTOR_WORDS] * SECTOR_WORDS + i % SECTOR_WORDS;  // 8 lanes share a 64-byte sector
      const uint64_t filter_word = filter[w];
      uint64_t bits[POSG][BITS];
#pragma unroll
      for (int g = 0; g < POSG; ++g) {
         // positions past the end are clamped (an in-bounds re-read) and not stored below
         const uint64_t* base = planes + static_cast<size_t>(min(pos_begin + g, last_pos)) * position_stride + w;
#pragma unroll
         for (int bit = 0; bit < BITS; ++bit) {
            bits[g][bit] = base[static_cast<size_t>(bit) * row_words];
         }
      }
#pragma unroll
      for (int g = 0; g < POSG; ++g) {
         if constexpr (KIND == KIND_ROWS) {
            acc[g][0] += static_cast<uint32_t>(__popcll(bits[g][0] & filter_word));
            continue;
         }
         constexpr int B1 = BITS > 1 ? 1 : 0;  // (one plane: never decoded)
         uint64_t low[4];
         low[0] = ~bits[g][B1] & ~bits[g][0] & filter_word;
         low[1] = ~bits[g][B1] & bits[g][0] & filter_word;
         low[2] = bits[g][B1] & ~bits[g][0] & filter_word;
         low[3] = bits[g][B1] & bits[g][0] & filter_word;
         uint64_t high[BITS <= 3 ? 2 : 8];
         if constexpr (BITS <= 2) {
            high[0] = ~0ull;  // the codes are the low pair
            high[1] = 0;
         } else if constexpr (BITS == 3) {
            high[0] = ~bits[g][2];
            high[1] = bits[g][2];
         } else if constexpr (BITS == 5) {
#pragma unroll
            for (int k = 0; k < 8; ++k) {
               high[k] = ((k & 1) != 0 ? bits[g][2] : ~bits[g][2]) & ((k & 2) != 0 ? bits[g][3] : ~bits[g][3]) &
                         ((k & 4) != 0 ? bits[g][4] : ~bits[g][4]);
            }
         }
#pragma unroll
         for (int symbol = 0; symbol < NSYM; ++symbol) {
            const uint32_t code = static_cast<uint32_t>(symbol) + 1u;
            acc[g][symbol] += static_cast<uint32_t>(__popcll(BITS <= 2 ? low[code & 3u] : (low[code & 3u] & high[code >> 2])));
         }
      }
   }
#pragma unroll
   for (int g = 0; g < POSG; ++g) {
#pragma unroll
      for (int symbol = 0; symbol < NSYM; ++symbol) {
         const uint32_t total = waveSumToLane63(acc[g][symbol]);
         if (lane == 63u && total != 0 && pos_begin + g < n_positions) {
            if constexpr (KIND == KIND_ROWS) {  // row -> its (position, symbol) counter
               const uint32_t target = reinterpret_cast<const uint32_t*>(batch.code_map[range])[pos_begin + g] - batch.target_base[range];
               atomicAdd(&batch.counts[range][q][target], total);
            } else if constexpr (KIND == KIND_MAPPED) {  // code -> the symbol it stands for at this position
               const uint32_t mapped = batch.code_map[range][static_cast<size_t>(pos_begin + g) * CODE_MAP_STRIDE + 1 + symbol];
               if (mapped < batch.out_symbols) {
                  atomicAdd(&batch.counts[range][q][static_cast<size_t>(pos_begin + g) * batch.out_symbols + mapped], total);
               }
            } else {
               atomicAdd(&batch.counts[range][q][static_cast<size_t>(pos_begin + g) * NSYM + symbol], total);
            }
         }
      }
   }
}

// ------------------------------------------------------------------------------------------------
// The adaptive planes.  At almost every position of a real alignment ONE symbol has nearly every row, and where not, three
// symbols cover all but a handful (the reference symbol, the gap or a lineage's substitution, one more), so a finalized store
// does not keep the ceil(log2(|valid| + 1)) code planes of the build (3 nucleotide, 5 amino-acid) everywhere: finalize() picks,
// per POSITION, the cheapest of
//    one-hot rows: 1, 2 or 3 rows, row j = the rows of the position's j-th most frequent valid symbol,
//    2 planes: codes 1..3 = the three most frequent valid symbols of the position,
//    3 planes: codes 1..7 = the seven most frequent (amino acids only: for nucleotides that is the full set),
//    the full identity planes,
// where the rows whose valid symbol the position does not store become explicit keys ("escapes": position << 37 | scan
// symbol << 32 | sequence, sorted; a second copy slice-major for the scan's escape pass).  Cost model (chooseLayouts in
// layout_choice.h, on the host from the unfiltered totals), in bytes the Mutations scan has to move: rows x row bytes +
// KEY_COST_BYTES per escape, the 22-symbol decode of the full amino-acid planes weighted by what it costs in VALU time, and a
// charge per change of layout between neighbouring positions: a scan launch takes runs of ONE layout (one-hot positions of
// any number of rows are one layout: a run of rows), and a run of a few positions costs its blocks the filter tile and the
// pipeline ramp all over again.  The build-time planes are freed afterwards: at 10 M sequences the nucleotide store shrinks
// from 112 GB to 38 GB of plane rows (+ 37 GB for the missing-symbol plane) and every consumer — the scan, the
// sparse-filter gather, filter leaves, FastaAligned — reads the adaptive planes.
// ------------------------------------------------------------------------------------------------
using silo_gpu_layout::KEY_COST_BYTES;

/// Re-encodes the build-time planes of every position into its adaptive layout; rows without a code go, with an atomic
/// cursor per (position, symbol), into that counter's exactly sized slice of the key list (sorted afterwards).
template <int BITS>
__global__ __launch_bounds__(256) void k_encode_adaptive(
   const uint64_t* __restrict__ scan, uint32_t row_words, uint32_t n_scan, const uint8_t* __restrict__ code_map, const uint32_t* __restrict__ row_of,
   const uint32_t* __restrict__ escape_first, uint32_t* __restrict__ escape_cursor, uint64_t* __restrict__ planes, uint64_t* __restrict__ escapes
) {
   const uint32_t p = blockIdx.y;
   const uint32_t w = blockIdx.x * blockDim.x + threadIdx.x;
   if (w >= row_words) {
      return;
   }
   uint64_t bits[BITS];
   uint64_t valid = 0;
#pragma unroll
   for (int bit = 0; bit < BITS; ++bit) {
      bits[bit] = scan[(static_cast<size_t>(p) * BITS + bit) * row_words + w];
      valid |= bits[bit];
   }
   const uint8_t* map = code_map + static_cast<size_t>(p) * CODE_MAP_STRIDE;
   uint64_t* out = planes + static_cast<size_t>(row_of[p]) * row_words + w;
   if ((map[0] & LAYOUT_IDENTITY) != 0) {
#pragma unroll
      for (int bit = 0; bit < BITS; ++bit) {
         out[static_cast<size_t>(bit) * row_words] = bits[bit];
      }
      return;
   }
   const uint32_t out_bits = map[0] & LAYOUT_ROWS_MASK;  // 2 or 3 code planes, or 0..3 one-hot rows
   const bool one_hot = (map[0] & LAYOUT_ONE_HOT) != 0;
   uint64_t out_plane[3] = {0, 0, 0};
   uint64_t coded = 0;
   if ((map[0] & LAYOUT_IMPLICIT) != 0) {  // the rows of the derived symbol are stored nowhere
      const uint32_t full_code = map[IMPLICIT_SLOT] + 1u;
      uint64_t match = ~0ull;
#pragma unroll
      for (int bit = 0; bit < BITS; ++bit) {
         match &= ((full_code >> bit) & 1u) != 0 ? bits[bit] : ~bits[bit];
      }
      coded |= match;
   }
   for (uint32_t code = 1; code < (one_hot ? out_bits + 1u : (1u << out_bits)); ++code) {
      const uint32_t symbol = map[code];
      if (symbol == 0xFFu) {
         continue;
      }
      const uint32_t full_code = symbol + 1u;
      uint64_t match = ~0ull;
#pragma unroll
      for (int bit = 0; bit < BITS; ++bit) {
         match &= ((full_code >> bit) & 1u) != 0 ? bits[bit] : ~bits[bit];
      }
      coded |= match;
#pragma unroll
      for (uint32_t bit = 0; bit < 3; ++bit) {
         if (one_hot ? code == bit + 1u : ((code >> bit) & 1u) != 0) {
            out_plane[bit] |= match;
         }
      }
   }
   for (uint32_t bit = 0; bit < out_bits; ++bit) {
      out[static_cast<size_t>(bit) * row_words] = out_plane[bit];
   }
   uint64_t escaped = valid & ~coded;
   while (escaped != 0) {
      const uint32_t row_bit = static_cast<uint32_t>(__builtin_ctzll(escaped));
      escaped &= escaped - 1;
      uint32_t full_code = 0;
#pragma unroll
      for (int bit = 0; bit < BITS; ++bit) {
         full_code |= static_cast<uint32_t>((bits[bit] >> row_bit) & 1ull) << bit;
      }
      const size_t counter = static_cast<size_t>(p) * n_scan + (full_code - 1u);
      const uint32_t slot = escape_first[counter] + atomicAdd(escape_cursor + counter, 1u);
      escapes[slot] = (static_cast<uint64_t>(p) << 37) | (static_cast<uint64_t>(full_code - 1u) << 32) | (static_cast<uint64_t>(w) * 64u + row_bit);
   }
}

/// The rows the code planes do not carry: one key per (position, symbol, sequence); grid.y = filter.  A thread takes
/// ESCAPE_KEYS_PER_THREAD keys a block-width apart (their loads and the filter lookups behind them are in flight together).
constexpr uint32_t ESCAPE_KEYS_PER_THREAD = 4;
__global__ __launch_bounds__(256) void k_scan_escapes(
   const uint64_t* __restrict__ escapes, uint32_t n_escapes, const ScanBatchArgs batch, uint32_t pos_begin
) {
   const uint32_t q = blockIdx.y;  // every filter: dense scan and sparse-filter gather of a range both read the same planes
   const uint32_t lane = threadIdx.x & 63u;
   const uint32_t first = blockIdx.x * (256u * ESCAPE_KEYS_PER_THREAD) + threadIdx.x;
   uint64_t key[ESCAPE_KEYS_PER_THREAD];
   bool selected[ESCAPE_KEYS_PER_THREAD];
#pragma unroll
   for (uint32_t k = 0; k < ESCAPE_KEYS_PER_THREAD; ++k) {
      const uint32_t i = first + k * 256u;
      key[k] = i < n_escapes ? escapes[i] : 0;
   }
#pragma unroll
   for (uint32_t k = 0; k < ESCAPE_KEYS_PER_THREAD; ++k) {
      const uint32_t sequence = static_cast<uint32_t>(key[k]);
      selected[k] = first + k * 256u < n_escapes && ((batch.filters[q][sequence >> 6] >> (sequence & 63u)) & 1ull) != 0;
   }
#pragma unroll
   for (uint32_t k = 0; k < ESCAPE_KEYS_PER_THREAD; ++k) {
      bool pending = selected[k];
      // keys of one position sit together and share a few symbols: one atomic per distinct counter and wave, not per key
      const uint32_t counter = (static_cast<uint32_t>(key[k] >> 37) - pos_begin) * batch.out_symbols + (static_cast<uint32_t>(key[k] >> 32) & 31u);
      for (uint64_t open = __ballot(pending); open != 0; open = __ballot(pending)) {
         const uint32_t leader = static_cast<uint32_t>(__builtin_ctzll(open));
         const uint32_t leader_counter = __shfl(counter, leader);
         const uint64_t same = __ballot(pending && counter == leader_counter);
         if (lane == leader) {
            atomicAdd(&batch.counts[0][q][leader_counter], static_cast<uint32_t>(__popcll(same)));
         }
         if (counter == leader_counter) {
            pending = false;
         }
      }
   }
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
/// One launch for up to ESCAPE_MAX_RANGES position ranges (the 12 genes of an AminoAcidMutations query): grid =
/// (blocks per slice, slice x range, filter).  Block j of a (range, slice) takes the chunks j, j + gridDim.x, ... of that
/// slice's keys of the scanned positions; where those begin and end is read from the store's slice index on the device.
struct EscapeSliceArgs {
   const uint64_t* filters[SILO_GPU_MAX_SCAN_BATCH];
   uint32_t row_words;
   uint32_t n_slices;
   uint32_t out_symbols;
   uint32_t block_keys;  // keys per block (even)
   struct Range {
      const uint64_t* keys;          // slice-major keys of the store
      const uint32_t* slice_first;   // [n_slices][positions + 1]
      uint32_t positions;
      uint32_t pos_begin;
      uint32_t pos_end;
      uint32_t* counts[SILO_GPU_MAX_SCAN_BATCH];  // of the range's first position
   } ranges[ESCAPE_MAX_RANGES];
};

/// Workgroup barrier for data exchanged through LDS only: waits for the wave's LDS operations, NOT for its outstanding global
/// loads — __syncthreads() is also a fence and drains vmcnt(0), which would stall a block on the loads it has prefetched for
/// its next step at every barrier.
__device__ __forceinline__ void ldsBarrier() {
   asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

/// FILTERS = filters a block serves with ONE pass over its keys (1, 2, 4 or 8: a batch of 8 filters keeps 8 x 16 KiB of filter
/// slices in LDS and reads every key once, not once per filter); blockIdx.z = first filter / FILTERS.
///
/// Counting.  The keys of a slice are sorted by (position, symbol), so the counters a chunk of keys adds to lie in a narrow
/// window behind the chunk's first key: the block counts into a window of LDS counters per filter (one LDS atomic per selected
/// key, no wave-level bookkeeping) and then adds the window to the table with CONTIGUOUS atomics — 64 consecutive counters per
/// wave instruction, the shape the memory side takes at full rate; a lane per scattered counter, as the first version did, is
/// an order of magnitude slower per add (MI355X guide, "Global float atomics": access shape).  A key past the window (sparse
/// stretches of keys) goes to the table directly.
template <int FILTERS>
constexpr uint32_t escapeKeysInFlight() {  // per thread: fewer for a batch, whose windows are narrower
   return FILTERS >= 8 ? 4u : (FILTERS >= 4 ? 8u : 16u);
}
constexpr uint32_t ESCAPE_CHUNKS_PER_BLOCK = 4;  // at most: consecutive chunks of IN_FLIGHT x 1024 keys a block counts into ONE window
template <int FILTERS>
constexpr uint32_t escapeWindow() {  // LDS counters per filter: 48 KiB of them for 1-4 filters (two blocks per CU), 28 KiB for 8 (beside 128 KiB of filter slices)
   return FILTERS >= 8 ? 896u : 12288u / FILTERS;
}
template <int FILTERS>
constexpr uint32_t escapeLdsBytes() {
   return FILTERS * (ESCAPE_SLICE_WORDS32 + escapeWindow<FILTERS>()) * static_cast<uint32_t>(sizeof(uint32_t));
}

/// grid = (blocks per slice, slice x range, filters / FILTERS).  Block j of a (range, slice) takes the keys [j, j + 1) x
/// ESCAPE_CHUNKS_PER_BLOCK chunks of that slice's keys of the scanned positions (an even first index: 16-byte loads of two
/// keys per lane — 8-byte loads stream at 0.54-0.70 of their rate); where the slice's keys begin and end is read from the
/// store's slice index on the device.  No barrier between a block's chunks: its waves run on by themselves, one waits for
/// its keys while another counts; two blocks per CU (<= 64 VGPRs, 64 KiB of LDS) cover each other's first and last steps.
template <int FILTERS, bool AGGREGATE = true>
__global__ __launch_bounds__(ESCAPE_SLICE_THREADS, FILTERS <= 4 ? 8 : 4) void k_scan_escapes_sliced(const EscapeSliceArgs args, uint32_t n_filters) {
   constexpr uint32_t IN_FLIGHT = escapeKeysInFlight<FILTERS>();
   constexpr uint32_t CHUNK_KEYS = ESCAPE_SLICE_THREADS * IN_FLIGHT;
   constexpr uint32_t WINDOW = escapeWindow<FILTERS>();
   const uint32_t BLOCK_KEYS = args.block_keys;  // even; chosen by the launcher so that a block's keys mostly fall into its window
   extern __shared__ uint32_t s_filter[];  // [FILTERS][ESCAPE_SLICE_WORDS32], then the counters [FILTERS][WINDOW]
   uint32_t* s_count = s_filter + FILTERS * ESCAPE_SLICE_WORDS32;
   const uint32_t first_filter = blockIdx.z * FILTERS;
   const uint32_t slice = blockIdx.y % args.n_slices;
   const EscapeSliceArgs::Range& range = args.ranges[blockIdx.y / args.n_slices];
   const uint32_t* first = range.slice_first + static_cast<size_t>(slice) * (range.positions + 1u);
   const uint32_t key_begin = first[range.pos_begin];
   const uint32_t key_end = first[range.pos_end];
   const uint32_t block_begin = (key_begin & ~1u) + blockIdx.x * BLOCK_KEYS;
   if (block_begin >= key_end) {
      return;  // (uniform) no keys for this block
   }
   const uint32_t block_end = min(block_begin + BLOCK_KEYS, key_end);
   // the two keys that bound the block's window of counters (slice-major keys are recoded for this kernel: counter of the store
   // << 32 | sequence, counter = position * symbols + symbol); the loads are under way while the filter slices come in
   const uint64_t first_key = range.keys[max(block_begin, key_begin)];
   const uint64_t last_key = range.keys[block_end - 1u];
   uint64_t any_bit = 0;
#pragma unroll
   for (int f = 0; f < FILTERS; ++f) {  // this slice of every filter: 16 bytes per thread, zeros past the end of the row (and for a filter past the last)
      const uint32_t first_word = slice * (ESCAPE_SLICE_WORDS32 / 2u);
      const bool present = first_filter + f < n_filters;
      const uint64_t* filter = args.filters[present ? first_filter + f : first_filter];
#pragma unroll
      for (uint32_t j = 0; j < ESCAPE_SLICE_WORDS32 / 4u / ESCAPE_SLICE_THREADS; ++j) {
         const uint32_t chunk = j * ESCAPE_SLICE_THREADS + threadIdx.x;  // 16-byte chunk of the slice
         const uint32_t word = first_word + chunk * 2u;
         const ulonglong2 v = present && word < args.row_words ? *reinterpret_cast<const ulonglong2*>(filter + word) : make_ulonglong2(0, 0);
         *reinterpret_cast<ulonglong2*>(s_filter + f * ESCAPE_SLICE_WORDS32 + chunk * 4u) = v;
         any_bit |= v.x | v.y;
      }
   }
   for (uint32_t j = threadIdx.x; j < FILTERS * WINDOW; j += ESCAPE_SLICE_THREADS) {
      s_count[j] = 0;
   }
   if (__syncthreads_or(any_bit != 0 ? 1 : 0) == 0) {
      return;  // no row of this slice is selected: none of its keys counts
   }
   const uint32_t slice_first_row = slice << ESCAPE_SLICE_SHIFT;
   const uint32_t range_first = range.pos_begin * args.out_symbols;
   // the window: the counters from the position of the block's first key on; what it reaches of the block's last key's position
   const uint32_t window_first = static_cast<uint32_t>(first_key >> 32) / args.out_symbols * args.out_symbols - range_first;
   const uint32_t window_used = min(WINDOW, (static_cast<uint32_t>(last_key >> 32) / args.out_symbols + 1u) * args.out_symbols - range_first - window_first);
   for (uint32_t base = block_begin; base < block_end; base += CHUNK_KEYS) {  // uniform per block
      uint64_t key[IN_FLIGHT];
#pragma unroll
      for (uint32_t k = 0; k < IN_FLIGHT / 2u; ++k) {
         const uint32_t i = base + (k * ESCAPE_SLICE_THREADS + threadIdx.x) * 2u;
         ulonglong2 pair = make_ulonglong2(~0ull, ~0ull);
         if (i < block_end) {
            pair = loadPlane16<true>(range.keys + i);  // (the key before the slice's first and the key behind its last, read along, are masked out)
         }
         key[2 * k] = i >= key_begin && i < block_end ? pair.x : ~0ull;
         key[2 * k + 1] = i + 1u < block_end ? pair.y : ~0ull;
      }
#pragma unroll
      for (uint32_t k = 0; k < IN_FLIGHT; ++k) {
         const bool valid = key[k] != ~0ull;  // (no key is all ones: a store has fewer than 2^32 - 1 counters)
         const uint32_t local = valid ? static_cast<uint32_t>(key[k]) - slice_first_row : 0u;
         const uint32_t counter = valid ? static_cast<uint32_t>(key[k] >> 32) - range_first : 0xFFFFFFFFu;
         const uint32_t in_window = counter - window_first;
         // The lanes of a wave hold consecutive keys of the sorted list: the keys of one counter sit side by side, and 64 LDS atomics
         // on ONE address take 64 LDS cycles (identical addresses do not combine for atomics).  So a stretch of lanes with one
         // counter adds its selected keys with TWO atomics: with `below` = the selected lanes below a lane (v_mbcnt of the ballot:
         // two instructions), the stretch's first lane adds -below, its last lane +below + its own key; the sum is the number of
         // selected keys in between (uint32 wrap-around; the window is read after the barrier).
         const uint32_t previous = __builtin_amdgcn_update_dpp(0xFFFFFFFEu, counter, 0x138, 0xf, 0xf, false);  // wave_shr:1 (lane 0 keeps the old value)
         const uint32_t next = __builtin_amdgcn_update_dpp(0xFFFFFFFEu, counter, 0x130, 0xf, 0xf, false);      // wave_shl:1 (lane 63 keeps the old value)
         const bool head = valid && counter != previous;
         const bool tail = valid && counter != next;
#pragma unroll
         for (int f = 0; f < FILTERS; ++f) {
            const bool is_selected = valid && ((s_filter[f * ESCAPE_SLICE_WORDS32 + (local >> 5)] >> (local & 31u)) & 1u) != 0;
            uint32_t* __restrict__ window = s_count + f * WINDOW;
            uint32_t* __restrict__ table = range.counts[first_filter + f < n_filters ? first_filter + f : first_filter];
            if constexpr (!AGGREGATE) {  // (measurement: one LDS atomic per selected key)
               if (is_selected && in_window < WINDOW) {
                  atomicAdd(&window[in_window], 1u);
               } else if (is_selected) {
                  atomicAdd(&table[counter], 1u);
               }
               continue;
            }
            const uint64_t selected = __ballot(is_selected);
            const uint32_t below = __builtin_amdgcn_mbcnt_hi(static_cast<uint32_t>(selected >> 32), __builtin_amdgcn_mbcnt_lo(static_cast<uint32_t>(selected), 0u));
            const uint32_t upto = below + (is_selected ? 1u : 0u);
            if (in_window < WINDOW) {
               if (head && below != 0) {
                  atomicAdd(&window[in_window], 0u - below);
               }
               if (tail && upto != 0) {
                  atomicAdd(&window[in_window], upto);
               }
            } else if (valid) {  // a key past the window: straight to the table
               if (head && below != 0) {
                  atomicAdd(&table[counter], 0u - below);
               }
               if (tail && upto != 0) {
                  atomicAdd(&table[counter], upto);
               }
            }
         }
      }
   }
   // the window goes to the table: contiguous atomics, 64 consecutive counters per wave instruction
   ldsBarrier();
#pragma unroll
   for (int f = 0; f < FILTERS; ++f) {
      uint32_t* __restrict__ counts = range.counts[first_filter + f < n_filters ? first_filter + f : first_filter] + window_first;
      for (uint32_t j = threadIdx.x; j < window_used; j += ESCAPE_SLICE_THREADS) {
         const uint32_t value = s_count[f * WINDOW + j];
         if (value != 0) {
            atomicAdd(&counts[j], value);
         }
      }
   }
}

/// The slice-major keys as k_scan_escapes_sliced reads them: (position * n_scan + symbol) << 32 | sequence — the counter a key
/// adds to is a subtraction away, no shifts, no multiplication per key (the kernel is bound by its integer work per key).
__global__ void k_recode_sliced_keys(uint64_t* __restrict__ keys, uint32_t n_keys, uint32_t n_scan) {
   const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
   if (i < n_keys) {
      const uint64_t key = keys[i];
      const uint32_t counter = static_cast<uint32_t>(key >> 37) * n_scan + (static_cast<uint32_t>(key >> 32) & 31u);
      keys[i] = (static_cast<uint64_t>(counter) << 32) | (key & 0xFFFFFFFFull);
   }
}

/// Did the encoding pass of a two-pass build fill every (position, symbol) slice of the key list exactly?
__global__ void k_check_cursors(const uint32_t* __restrict__ first, const uint32_t* __restrict__ cursor, uint32_t n_counters, uint32_t* __restrict__ mismatches) {
   const uint32_t counter = blockIdx.x * blockDim.x + threadIdx.x;
   if (counter < n_counters && cursor[counter] != first[counter + 1] - first[counter]) {
      atomicAdd(mismatches, 1u);
   }
}

/// first[slice][p] = index of the first slice-major key of (slice, position >= p): a binary search per entry.
__global__ __launch_bounds__(256) void k_slice_index(
   const uint64_t* __restrict__ keys, uint32_t n_keys, uint32_t slice_shift, uint32_t n_slices, uint32_t positions, uint32_t* __restrict__ first
) {
   const uint32_t entry = blockIdx.x * blockDim.x + threadIdx.x;
   if (entry >= n_slices * (positions + 1u)) {
      return;
   }
   const uint32_t slice = entry / (positions + 1u);
   const uint32_t position = entry % (positions + 1u);
   uint32_t lo = 0, hi = n_keys;
   while (lo < hi) {  // keys before (slice, position): a smaller slice, or the same slice and a smaller position
      const uint32_t mid = lo + (hi - lo) / 2;
      const uint32_t key_slice = static_cast<uint32_t>(keys[mid]) >> slice_shift;
      const uint32_t key_position = static_cast<uint32_t>(keys[mid] >> 37);
      if (key_slice < slice || (key_slice == slice && key_position < position)) {
         lo = mid + 1;
      } else {
         hi = mid;
      }
   }
   first[entry] = lo;
}

// ------------------------------------------------------------------------------------------------
// Derived symbols (LAYOUT_IMPLICIT).  At almost every position of an alignment ONE symbol has nearly every row.  The reference
// leaves that symbol's bitmap out and rebuilds its count as |filter| - #missing - the other symbols' counts
// (position.cpp:102-127, mutations.cpp:74-95); the dense restatement of the same idea: such a position stores NO row for that
// symbol, and a scan
//   1. counts the other valid symbols as ever (their one-hot rows, their escape keys) — into PRIVATE tables in scratch,
//   2. counts, per position, the rows of the filter that have no valid symbol there: those inside a run of the missing symbol
//      (k_scan_missing_runs: +1 where a selected row's run starts, -1 where it ends, summed along the positions afterwards)
//      and those with an ambiguity code (k_count_sparse_keys),
//   3. k_finish_scan: derived count = |filter| - (2.) - sum of (1.) at the position; private tables -> the caller's.
// The filter's cardinality comes from the prepare step (k_compact_filter, counter [2]).
// ------------------------------------------------------------------------------------------------
constexpr uint32_t DERIVED_MAX_RANGES = 16;
constexpr uint32_t DERIVED_THREADS = 1024;
constexpr uint32_t SPARSE_KEYS_PER_THREAD = 4;
constexpr uint32_t RUNS_IN_FLIGHT = 4;        // runs per thread whose loads are in flight together (k_scan_missing_runs)

/// A range of a scan with derived symbols.  Its private tables: per filter `stride` words of scratch — counts[n][n_scan], then
/// diff[n + 1] (selected rows entering / leaving a run of the missing symbol at each position), then ambiguous[n].
struct DerivedRange {
   uint32_t* scratch;        // of filter 0
   uint32_t stride;          // words per filter
   uint32_t n_positions;
   uint32_t n_scan;
   uint32_t pos_begin;
   const uint8_t* code_map;  // of the store's position 0; nullptr: no position of this store derives a symbol
   const uint64_t* run_keys;
   const uint32_t* run_ends;
   const uint32_t* run_slice_first;  // [n_run_slices + 1]
   const uint64_t* sparse_keys;      // position << 37 | symbol << 32 | sequence, ascending
   uint32_t sparse_begin;            // the keys of the range's positions
   uint32_t sparse_end;
   uint32_t* caller_counts[SILO_GPU_MAX_SCAN_BATCH];  // at the range's first position
};
struct DerivedArgs {
   const uint64_t* filters[SILO_GPU_MAX_SCAN_BATCH];
   const uint32_t* counters;  // of the prepare step: [q * SPARSE_COUNTER_STRIDE + 2] = the cardinality of filter q
   uint32_t row_words;
   uint32_t n_run_slices;
   uint32_t n_ranges;
   uint32_t first_unit[DERIVED_MAX_RANGES + 1];  // blocks per range (k_count_sparse_keys, k_finish_scan: each their own)
   DerivedRange ranges[DERIVED_MAX_RANGES];
};

/// grid = (blocks per slice, slice of 2^17 sequences x range, filter).  The block keeps its slice of the filter in LDS (16 KiB) and, where it
/// fits (LDS_DIFF), the diff of the whole range as well (<= ~140 KiB: 35 000 positions), so that the adds of a slice's runs —
/// two per selected run — are LDS atomics and only the non-zero entries go to memory (256 contiguous bytes per wave instruction).
template <bool LDS_DIFF>
__global__ __launch_bounds__(DERIVED_THREADS) void k_scan_missing_runs(const DerivedArgs args) {
   extern __shared__ uint32_t s_runs[];  // [ESCAPE_SLICE_WORDS32] the filter slice, then [n + 1] the diff
   uint32_t* s_diff = s_runs + ESCAPE_SLICE_WORDS32;
   const uint32_t q = blockIdx.z;
   const uint32_t slice = blockIdx.y % args.n_run_slices;
   const DerivedRange& range = args.ranges[blockIdx.y / args.n_run_slices];
   if (range.code_map == nullptr) {
      return;  // (uniform) nothing is derived in this store
   }
   const uint32_t run_begin = range.run_slice_first[slice];
   const uint32_t run_end = range.run_slice_first[slice + 1];
   if (run_begin + blockIdx.x * (DERIVED_THREADS * RUNS_IN_FLIGHT) >= run_end) {
      return;  // (uniform) no chunk of runs for this block
   }
   uint64_t any_bit = 0;
   {
      const uint64_t* filter = args.filters[q];
      const uint32_t first_word = slice * (ESCAPE_SLICE_WORDS32 / 2u);
#pragma unroll
      for (uint32_t j = 0; j < ESCAPE_SLICE_WORDS32 / 4u / DERIVED_THREADS; ++j) {
         const uint32_t chunk = j * DERIVED_THREADS + threadIdx.x;  // 16-byte chunk of the slice
         const uint32_t word = first_word + chunk * 2u;
         const ulonglong2 v = word < args.row_words ? *reinterpret_cast<const ulonglong2*>(filter + word) : make_ulonglong2(0, 0);
         *reinterpret_cast<ulonglong2*>(s_runs + chunk * 4u) = v;
         any_bit |= v.x | v.y;
      }
   }
   const uint32_t n = range.n_positions;
   if constexpr (LDS_DIFF) {
      for (uint32_t j = threadIdx.x; j <= n; j += DERIVED_THREADS) {
         s_diff[j] = 0;
      }
   }
   if (__syncthreads_or(any_bit != 0 ? 1 : 0) == 0) {
      return;  // no row of this slice is selected
   }
   uint32_t* __restrict__ diff = range.scratch + static_cast<size_t>(q) * range.stride + static_cast<size_t>(n) * range.n_scan;
   const uint32_t slice_first_row = slice << ESCAPE_SLICE_SHIFT;
   const uint32_t pos_end = range.pos_begin + n;
   // the slice's runs are dealt to the gridDim.x blocks of the slice in chunks of RUNS_IN_FLIGHT x 1024 (loads of a chunk in flight together)
   for (uint32_t base = run_begin + blockIdx.x * (DERIVED_THREADS * RUNS_IN_FLIGHT); base < run_end; base += gridDim.x * (DERIVED_THREADS * RUNS_IN_FLIGHT)) {
      uint64_t key[RUNS_IN_FLIGHT];
      uint32_t run_last[RUNS_IN_FLIGHT];
#pragma unroll
      for (uint32_t k = 0; k < RUNS_IN_FLIGHT; ++k) {
         const uint32_t i = base + k * DERIVED_THREADS + threadIdx.x;
         key[k] = i < run_end ? range.run_keys[i] : 0;
         run_last[k] = i < run_end ? range.run_ends[i] : 0;  // (an empty run: start >= end below)
      }
#pragma unroll
      for (uint32_t k = 0; k < RUNS_IN_FLIGHT; ++k) {
         const uint32_t local = (static_cast<uint32_t>(key[k] >> 32) - slice_first_row) & ((1u << ESCAPE_SLICE_SHIFT) - 1u);
         const bool selected = ((s_runs[local >> 5] >> (local & 31u)) & 1u) != 0;
         const uint32_t start = max(static_cast<uint32_t>(key[k]), range.pos_begin);
         const uint32_t end = min(run_last[k], pos_end);
         if (selected && start < end) {
            if constexpr (LDS_DIFF) {
               atomicAdd(&s_diff[start - range.pos_begin], 1u);
               atomicAdd(&s_diff[end - range.pos_begin], 0xFFFFFFFFu);
            } else {
               atomicAdd(&diff[start - range.pos_begin], 1u);
               atomicAdd(&diff[end - range.pos_begin], 0xFFFFFFFFu);
            }
         }
      }
   }
   if constexpr (LDS_DIFF) {
      __syncthreads();
      for (uint32_t j = threadIdx.x; j <= n; j += DERIVED_THREADS) {
         const uint32_t value = s_diff[j];
         if (value != 0) {
            atomicAdd(&diff[j], value);
         }
      }
   }
}

/// ambiguous[p] += the rows of filter blockIdx.y among the sparse keys (ambiguity codes) of position p: one global filter
/// lookup per key (these are ~1e-5 of the cells), one atomic per distinct position and wave.
__global__ __launch_bounds__(256) void k_count_sparse_keys(const DerivedArgs args) {
   const uint32_t q = blockIdx.y;
   const uint32_t lane = threadIdx.x & 63u;
   uint32_t r = 0;
   while (r + 1 < args.n_ranges && blockIdx.x >= args.first_unit[r + 1]) {
      ++r;
   }
   const DerivedRange& range = args.ranges[r];
   const uint32_t n = range.n_positions;
   uint32_t* __restrict__ ambiguous = range.scratch + static_cast<size_t>(q) * range.stride + static_cast<size_t>(n) * range.n_scan + n + 1u;
   const uint32_t first = range.sparse_begin + (blockIdx.x - args.first_unit[r]) * (256u * SPARSE_KEYS_PER_THREAD) + threadIdx.x;
   uint64_t key[SPARSE_KEYS_PER_THREAD];
#pragma unroll
   for (uint32_t k = 0; k < SPARSE_KEYS_PER_THREAD; ++k) {
      const uint32_t i = first + k * 256u;
      key[k] = i < range.sparse_end ? range.sparse_keys[i] : 0;
   }
#pragma unroll
   for (uint32_t k = 0; k < SPARSE_KEYS_PER_THREAD; ++k) {
      const uint32_t sequence = static_cast<uint32_t>(key[k]);
      bool pending = first + k * 256u < range.sparse_end && ((args.filters[q][sequence >> 6] >> (sequence & 63u)) & 1ull) != 0;
      const uint32_t counter = static_cast<uint32_t>(key[k] >> 37) - range.pos_begin;
      for (uint64_t open = __ballot(pending); open != 0; open = __ballot(pending)) {
         const uint32_t leader = static_cast<uint32_t>(__builtin_ctzll(open));
         const uint32_t leader_counter = __shfl(counter, leader);
         const uint64_t same = __ballot(pending && counter == leader_counter);
         if (lane == leader) {
            atomicAdd(&ambiguous[leader_counter], static_cast<uint32_t>(__popcll(same)));
         }
         if (counter == leader_counter) {
            pending = false;
         }
      }
   }
}

/// The last step of a scan with derived symbols: grid = (blocks of 1024 positions dealt to the ranges, filter).  A thread
/// owns a position: the rows of the filter inside a run of the missing symbol there (the sum of diff up to it: the part
/// before the block's positions summed by the block itself, then a scan over the block), plus those with an ambiguity code,
/// are the rows without a valid symbol; what is left of the filter after them and after the other symbols' counts is the
/// derived symbol's count.  The private table is added to the caller's.
__global__ __launch_bounds__(DERIVED_THREADS) void k_finish_scan(const DerivedArgs args) {
   __shared__ uint32_t s_before[DERIVED_THREADS / 64];
   __shared__ uint32_t s_own[DERIVED_THREADS / 64];
   const uint32_t q = blockIdx.y;
   const uint32_t lane = threadIdx.x & 63u;
   const uint32_t wave = threadIdx.x >> 6;
   uint32_t r = 0;
   while (r + 1 < args.n_ranges && blockIdx.x >= args.first_unit[r + 1]) {
      ++r;
   }
   const DerivedRange& range = args.ranges[r];
   const uint32_t n = range.n_positions;
   const uint32_t n_scan = range.n_scan;
   const uint32_t first_position = (blockIdx.x - args.first_unit[r]) * DERIVED_THREADS;
   const uint32_t p = first_position + threadIdx.x;
   const uint32_t* __restrict__ counts = range.scratch + static_cast<size_t>(q) * range.stride;
   const uint32_t* __restrict__ diff = counts + static_cast<size_t>(n) * n_scan;
   const uint32_t* __restrict__ ambiguous = diff + n + 1u;
   uint32_t without_symbol = 0;  // rows of the filter that have no valid symbol at p
   if (range.code_map != nullptr) {  // (uniform)
      uint32_t before = 0;
      for (uint32_t j = threadIdx.x; j < first_position; j += DERIVED_THREADS) {
         before += diff[j];
      }
      const uint32_t scanned = waveSumToLane63(p < n ? diff[p] : 0u);  // inclusive over the wave
      before = waveSumToLane63(before);
      if (lane == 63u) {
         s_before[wave] = before;
         s_own[wave] = scanned;
      }
      __syncthreads();
      without_symbol = scanned;
      for (uint32_t k = 0; k < DERIVED_THREADS / 64; ++k) {
         without_symbol += s_before[k] + (k < wave ? s_own[k] : 0u);
      }
      if (p < n) {
         without_symbol += ambiguous[p];
      }
   }
   if (p >= n) {
      return;
   }
   uint32_t* __restrict__ out = range.caller_counts[q] + static_cast<size_t>(p) * n_scan;
   const uint32_t* __restrict__ cell = counts + static_cast<size_t>(p) * n_scan;
   uint32_t others = 0;
   for (uint32_t symbol = 0; symbol < n_scan; ++symbol) {
      const uint32_t count = cell[symbol];
      others += count;
      if (count != 0) {
         out[symbol] += count;  // scans of one table are ordered on a stream: no atomic needed
      }
   }
   if (range.code_map != nullptr) {
      const uint8_t* map = range.code_map + static_cast<size_t>(range.pos_begin + p) * CODE_MAP_STRIDE;
      if ((map[0] & LAYOUT_IMPLICIT) != 0) {
         const uint32_t derived = args.counters[q * SPARSE_COUNTER_STRIDE + 2] - without_symbol - others;
         if (derived != 0) {
            out[map[IMPLICIT_SLOT]] += derived;
         }
      }
   }
}

// ------------------------------------------------------------------------------------------------
// Cardinalities are accumulated into SILO_GPU_COUNT_SHARDS 64-bit counters (shard = block % shards):
// thousands of waves adding to ONE word serialise at ~12 ns per atomic (the guide's "dequeue" row);
// spreading them over 64 words removes that tail.  The host sums the shards.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ void addToCountShard(unsigned long long* shards, uint32_t wave_total_lane63) {
   if ((threadIdx.x & 63u) == 63u && wave_total_lane63 != 0) {
      atomicAdd(shards + (blockIdx.x % SILO_GPU_COUNT_SHARDS), static_cast<unsigned long long>(wave_total_lane63));
   }
}


// ------------------------------------------------------------------------------------------------
// K2: popcount of one row-sized bitset
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_popcount(
   const uint64_t* __restrict__ bitset, uint32_t row_words, unsigned long long* __restrict__ out_shards
) {
   const uint32_t n_chunks = row_words / 2;
   uint32_t acc = 0;
   for (uint32_t chunk = blockIdx.x * blockDim.x + threadIdx.x; chunk < n_chunks; chunk += gridDim.x * blockDim.x) {
      const ulonglong2 v = *reinterpret_cast<const ulonglong2*>(bitset + 2 * chunk);
      acc += static_cast<uint32_t>(__popcll(v.x)) + static_cast<uint32_t>(__popcll(v.y));
   }
   addToCountShard(out_shards, waveSumToLane63(acc));
}

// ------------------------------------------------------------------------------------------------
// K3: fused filter evaluator.  One 64-lane wave per block; each lane owns TWO bitset words (one 16-byte
// access per leaf).  The program sits in the kernel-argument segment and is fetched with scalar loads
// (uniform control flow).  Slots live in LDS as [slot][lane] (16 B per lane: conflict-free b128
// accesses), typically a handful -> full occupancy; leaves are never staged: the n-ary instructions
// stream them from HBM 8 independent loads at a time, single leaf operands are loaded on use.
// ------------------------------------------------------------------------------------------------
constexpr int EVAL_THREADS = 64;
constexpr uint32_t EVAL_WORDS_PER_BLOCK = EVAL_THREADS * 2;

struct FilterEvalArgs {
   uint32_t n_instructions;
   uint32_t sequence_count;
   uint32_t row_words;
   uint32_t n_slots;
   uint64_t* out;
   unsigned long long* out_count;
   uint32_t* ticket;                 // count slot: blocks done so far
   unsigned long long* host_total;   // count slot: page-locked host word the last block stores the total into
   const uint64_t* leaves[SILO_GPU_MAX_LEAVES];
   uint32_t code[2 * SILO_GPU_MAX_INSTRUCTIONS];
};

// Tried and dropped (profiles/r01_k3_variants.md): 16 instead of 8 leaf loads in flight per n-ary instruction, and
// fetching the first 24 leaves up front into LDS (register-staged: spilled to scratch; LDS-DMA global_load_lds_dwordx4:
// no spill) — neither moved the kernel time of the 32-column program (19-21 us at 10 M sequences either way).  Round 3:
// blocks of 4 waves that fetch ALL leaves of a 128-word tile into LDS at once (32 loads in flight per tile) before wave 0
// evaluates: 30 us instead of 20 (profiles/r03_notes.md) — the kernel is not waiting for its loads.
/// The end of a filter kernel's wave (64 lanes, one per pair of result words): the popcount of the result goes to the count
/// shards, and with a count slot the last block hands the total to the host.
__device__ __forceinline__ void deliverFilterCount(const FilterEvalArgs& args, silo_gpu::Word2 result, uint32_t lane) {
   if (args.out_count != nullptr && args.ticket == nullptr) {
      const uint32_t bits = static_cast<uint32_t>(__popcll(result.x)) + static_cast<uint32_t>(__popcll(result.y));
      addToCountShard(args.out_count, waveSumToLane63(bits));
   }
   if (args.ticket != nullptr) {
      // Count slot: the last block to get here sums the shards, hands the total to the host through page-locked
      // memory (no copy, no stream synchronisation on the host side) and re-arms shards and tickets for the next
      // launch.  Atomics on one word serialise at ~12 ns each, so "last" is found in two levels: a ticket per shard
      // class (blocks b with b % 64 == c), and a main ticket taken by the block that completes its class.
      // Ordering uses only the atomics themselves (all performed at device scope, i.e. at the memory side): the shard
      // add is a RETURNING atomic, so it has been performed when its result arrives, and the ticket is taken after
      // that.  A __threadfence() here would be a release fence = an L2 write-back per block (the L2s of the 8 XCDs are
      // not coherent with each other), which doubled the kernel time when tried.
      const uint32_t bits = static_cast<uint32_t>(__popcll(result.x)) + static_cast<uint32_t>(__popcll(result.y));
      const uint32_t wave_total = waveSumToLane63(bits);
      uint32_t last = 0;
      if (lane == 63) {
         const uint32_t shard_class = blockIdx.x % SILO_GPU_COUNT_SHARDS;
         unsigned long long before = 0;
         if (wave_total != 0) {
            before = atomicAdd(args.out_count + shard_class, static_cast<unsigned long long>(wave_total));
         }
         // the ticket increment is made to depend on the value the shard add returned
         const uint32_t one = 1u + static_cast<uint32_t>((before >> 63) & 1ull);  // a shard never reaches 2^63: always 1
         const uint32_t blocks_in_class = (gridDim.x - 1 - shard_class) / SILO_GPU_COUNT_SHARDS + 1;
         const uint32_t class_ticket = atomicAdd(args.ticket + 1 + shard_class, one);
         if (class_ticket == blocks_in_class - 1) {
            const uint32_t classes = min(gridDim.x, static_cast<uint32_t>(SILO_GPU_COUNT_SHARDS));
            const uint32_t cleared = atomicExch(args.ticket + 1 + shard_class, 0u);
            last = atomicAdd(args.ticket, 1u + (cleared >> 31)) == classes - 1 ? 1u : 0u;
         }
      }
      last = __shfl(last, 63);
      if (last != 0) {
         const unsigned long long shard = atomicExch(args.out_count + lane, 0ull);  // EVAL_THREADS == SILO_GPU_COUNT_SHARDS
         const uint32_t total = waveSumToLane63(static_cast<uint32_t>(shard));      // a cardinality fits 32 bits
         if (lane == 63) {
            atomicExch(args.ticket, 0u);
            __hip_atomic_store(args.host_total, static_cast<unsigned long long>(total), __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
         }
      }
   }
}

template <uint32_t BATCH>
__global__ __launch_bounds__(EVAL_THREADS) void k_filter_eval(const FilterEvalArgs args) {
   extern __shared__ ulonglong2 s_slots[];  // [n_slots][EVAL_THREADS]
   using silo_gpu::Word2;
   const uint32_t lane = threadIdx.x;
   const uint32_t w = (blockIdx.x * EVAL_THREADS + lane) * 2;  // row_words is even (multiple of 32)
   const bool active = w < args.row_words;
   const uint32_t w_safe = active ? w : 0;
   Word2 valid{0, 0};
   if (active) {
      valid = {silo_gpu::valid_mask(w, args.sequence_count), silo_gpu::valid_mask(w + 1, args.sequence_count)};
   }
   const auto leaf = [&](uint32_t index) -> Word2 {
      const ulonglong2 v = *reinterpret_cast<const ulonglong2*>(args.leaves[index] + w_safe);
      return {v.x, v.y};
   };
   const auto get = [&](uint32_t index) -> Word2 {
      if (index >= SILO_GPU_LEAF_OPERAND) {
         return leaf(index - SILO_GPU_LEAF_OPERAND);
      }
      const ulonglong2 v = s_slots[index * EVAL_THREADS + lane];
      return {v.x, v.y};
   };
   const auto set = [&](uint32_t index, Word2 value) { s_slots[index * EVAL_THREADS + lane] = make_ulonglong2(value.x, value.y); };

   Word2 result = silo_gpu::bitprog_run<Word2, BATCH>(args.code, args.n_instructions, valid, get, set, leaf);
   result = result & valid;
   if (active && args.out != nullptr) {
      *reinterpret_cast<ulonglong2*>(args.out + w) = make_ulonglong2(result.x, result.y);
   }
   deliverFilterCount(args, result, lane);
}

// ------------------------------------------------------------------------------------------------
// K3b: the fused filter evaluator for a BATCH of programs — the filter -> Aggregated queries that are in flight at the
// same time (silo_api runs one request thread each; intersection.cpp:111-126, union.cpp:39-44 and threshold.cpp:93-128
// then run once per request).  One query of 32 columns is 40 MB at 10 M sequences: 5 us of HBM time, the same order as a
// launch, so a query on its own is latency-bound (k_filter_eval: ~1 wave per SIMD).  Q programs in ONE launch put
// Q x 1221 waves on the chip and stream at memory speed.  blockIdx.x = program (fastest: programs that name the same
// plane read the same tile of it close in time, so it is served from L2 / Infinity Cache), blockIdx.y = column tile of
// EVAL_BATCH_THREADS * 2 words; every wave works on its own 128 words (no barrier).  The programs do not fit the
// kernel-argument segment, so they sit in a device table (headers, then code and leaf pointers per program) and are
// fetched with scalar loads; counts go to EVAL_BATCH_SHARDS counters per program.
// ------------------------------------------------------------------------------------------------
constexpr int EVAL_BATCH_THREADS = 256;
constexpr uint32_t EVAL_BATCH_SHARDS = 16;

struct BatchProgramHeader {
   uint32_t n_instructions;
   uint32_t n_slots;
   uint32_t code_offset;  // bytes from the start of the table, 2 * n_instructions uint32
   uint32_t leaf_offset;  // bytes from the start of the table, n_leaves device pointers
   uint64_t* out;         // bitset of the result (row_words words) or nullptr
   uint64_t reserved;
};

__global__ __launch_bounds__(EVAL_BATCH_THREADS) void k_filter_eval_batch(
   const uint8_t* __restrict__ table, uint32_t first_program, uint32_t sequence_count, uint32_t row_words, uint32_t max_slots,
   uint32_t* __restrict__ counts
) {
   extern __shared__ ulonglong2 s_slots[];  // [wave][max_slots][64]
   using silo_gpu::Word2;
   const uint32_t program = first_program + blockIdx.x;
   const BatchProgramHeader header = reinterpret_cast<const BatchProgramHeader*>(table)[program];
   const uint32_t* __restrict__ code = reinterpret_cast<const uint32_t*>(table + header.code_offset);
   const uint64_t* const* __restrict__ leaves = reinterpret_cast<const uint64_t* const*>(table + header.leaf_offset);
   const uint32_t lane = threadIdx.x & 63u;
   ulonglong2* slots = s_slots + static_cast<size_t>(threadIdx.x >> 6) * max_slots * 64u;
   const uint32_t w = (blockIdx.y * EVAL_BATCH_THREADS + threadIdx.x) * 2;  // row_words is even (multiple of 32)
   const bool active = w < row_words;
   const uint32_t w_safe = active ? w : 0;
   Word2 valid{0, 0};
   if (active) {
      valid = {silo_gpu::valid_mask(w, sequence_count), silo_gpu::valid_mask(w + 1, sequence_count)};
   }
   const auto leaf = [&](uint32_t index) -> Word2 {
      const ulonglong2 v = loadGlobal16(leaves[index] + w_safe);
      return {v.x, v.y};
   };
   const auto get = [&](uint32_t index) -> Word2 {
      if (index >= SILO_GPU_LEAF_OPERAND) {
         return leaf(index - SILO_GPU_LEAF_OPERAND);
      }
      const ulonglong2 v = slots[index * 64u + lane];
      return {v.x, v.y};
   };
   const auto set = [&](uint32_t index, Word2 value) { slots[index * 64u + lane] = make_ulonglong2(value.x, value.y); };

   // (16 leaf loads in flight and non-temporal leaf loads were tried: +2 % and +1 %, within the noise — profiles/r02_filter_batch.md)
   Word2 result = silo_gpu::bitprog_run<Word2, 8>(code, header.n_instructions, valid, get, set, leaf);
   result = result & valid;
   if (active && header.out != nullptr) {
      *reinterpret_cast<ulonglong2*>(header.out + w) = make_ulonglong2(result.x, result.y);
   }
   const uint32_t bits = static_cast<uint32_t>(__popcll(result.x)) + static_cast<uint32_t>(__popcll(result.y));
   const uint32_t wave_total = waveSumToLane63(bits);
   if (lane == 63u && wave_total != 0) {
      atomicAdd(counts + program * EVAL_BATCH_SHARDS + ((blockIdx.y * (EVAL_BATCH_THREADS / 64) + (threadIdx.x >> 6)) % EVAL_BATCH_SHARDS), wave_total);
   }
}

// ------------------------------------------------------------------------------------------------
// plane writers shared by B1 / B2: `symbol` is this lane's symbol for sequence 64*word+lane
// (SILO_GPU_SYMBOL_NONE contributes no bit).
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ void emitWord(
   const SeqStoreDev& store, uint32_t position, uint32_t word, uint32_t symbol, bool whole_word,
   uint64_t* sparse, uint32_t* sparse_count, uint32_t sparse_capacity
) {
   const uint32_t lane = threadIdx.x & 63u;
   // valid mutation symbols: the bits of their code go to the bit-sliced scan planes
   const bool is_scan = symbol < store.n_symbols && store.kind[symbol] == PLANE_SCAN;
   const auto put = [&](uint64_t* dst, uint64_t mask) {  // one word of a plane row, by lane 0
      if (mask != 0 && lane == 0) {
         if (whole_word) {
            *dst = mask;
         } else {
            atomicOr(reinterpret_cast<unsigned long long*>(dst), static_cast<unsigned long long>(mask));
         }
      }
   };
   if (store.build_mode == BUILD_COUNT) {  // first pass of a two-pass build: how many rows have which valid symbol here
      const uint32_t scan_index = is_scan ? store.index[symbol] : 0xFFu;
      // ... and how many sparsely stored symbols there are in all (none is stored: the second pass gets a buffer that holds them)
      const uint64_t sparse_lanes = __ballot(symbol < store.n_symbols && store.kind[symbol] == PLANE_SPARSE);
      if (sparse_lanes != 0 && lane == 0) {
         atomicAdd(sparse_count, static_cast<uint32_t>(__popcll(sparse_lanes)));
      }
      for (uint64_t remaining = __ballot(is_scan); remaining != 0;) {
         const uint32_t leader = static_cast<uint32_t>(__builtin_ctzll(remaining));
         const uint32_t leader_index = __shfl(scan_index, leader);
         const uint64_t same = __ballot(is_scan && scan_index == leader_index);
         if (lane == leader) {
            atomicAdd(store.enc_counts + static_cast<size_t>(position) * store.n_scan + leader_index, static_cast<uint32_t>(__popcll(same)));
         }
         remaining &= ~same;
      }
      return;
   }
   if (store.build_mode == BUILD_ENCODE) {  // second pass: straight into the position's adaptive layout
      const uint8_t* map = store.enc_code_map + static_cast<size_t>(position) * CODE_MAP_STRIDE;
      const uint32_t rows_here = map[0] & LAYOUT_ROWS_MASK;
      const bool identity = (map[0] & LAYOUT_IDENTITY) != 0;
      const bool one_hot = (map[0] & LAYOUT_ONE_HOT) != 0;
      const uint32_t scan_index = is_scan ? store.index[symbol] : 0xFFu;
      // the position's derived symbol is stored nowhere: no row, no key
      const bool derived = is_scan && (map[0] & LAYOUT_IMPLICIT) != 0 && map[IMPLICIT_SLOT] == scan_index;
      uint32_t code = 0;  // the code (or 1 + one-hot row) of this lane's symbol here, 0 = not stored
      if (is_scan && !derived) {
         if (identity) {
            code = scan_index + 1u;
         } else {
            const uint32_t n_codes = one_hot ? rows_here + 1u : (1u << rows_here);
            for (uint32_t candidate = 1; candidate < n_codes; ++candidate) {
               code = map[candidate] == scan_index ? candidate : code;
            }
         }
      }
      uint64_t* rows = store.enc_planes + static_cast<size_t>(store.enc_row_of[position]) * store.row_words + word;
      for (uint32_t row = 0; row < rows_here; ++row) {
         put(rows + static_cast<size_t>(row) * store.row_words, __ballot(one_hot ? code == row + 1u : ((code >> row) & 1u) != 0));
      }
      if (is_scan && code == 0 && !derived) {  // a valid symbol the position does not store: an escape key in the symbol's slice of the list
         const size_t counter = static_cast<size_t>(position) * store.n_scan + scan_index;
         const uint32_t slot = store.enc_first[counter] + atomicAdd(store.enc_cursor + counter, 1u);
         if (slot < store.enc_first[counter + 1]) {  // (more rows than the first pass counted: dropped, the cursor tells)
            store.enc_escapes[slot] = (static_cast<uint64_t>(position) << 37) | (static_cast<uint64_t>(scan_index) << 32) | (static_cast<uint64_t>(word) * 64u + lane);
         }
      }
   } else {
      const uint32_t code = is_scan ? static_cast<uint32_t>(store.index[symbol]) + 1u : 0u;
      uint64_t* scan_word = store.scan + static_cast<size_t>(position) * store.n_bits * store.row_words + word;
      for (uint32_t bit = 0; bit < store.n_bits; ++bit) {
         put(scan_word + static_cast<size_t>(bit) * store.row_words, __ballot(((code >> bit) & 1u) != 0));
      }
   }
   // every other symbol: its own plane (extra) or the sorted key list (sparse)
   for (uint32_t s = 0; s < store.n_symbols; ++s) {
      const uint8_t kind = store.kind[s];
      if (kind == PLANE_SCAN) {
         continue;
      }
      const uint64_t mask = __ballot(symbol == s);
      if (mask == 0) {
         continue;
      }
      if (kind == PLANE_EXTRA) {
         if (lane == 0 && store.runs_at_build == 0) {  // (two-pass build: the missing symbol goes to its runs, the kernels track them)
            uint64_t* dst = planePtr(store, position, s) + word;
            if (whole_word) {
               *dst = mask;
            } else {
               atomicOr(reinterpret_cast<unsigned long long*>(dst), static_cast<unsigned long long>(mask));
            }
         }
      } else if (symbol == s) {
         const uint32_t slot = atomicAdd(sparse_count, 1u);
         if (slot < sparse_capacity) {
            sparse[slot] = (static_cast<uint64_t>(position) << 37) | (static_cast<uint64_t>(s) << 32) |
                           (static_cast<uint64_t>(word) * 64u + lane);
         }
      }
   }
}

// ------------------------------------------------------------------------------------------------
// Runs of the missing symbol while a store is built in two passes (SeqStoreDev::runs_at_build): a lane of the build kernels
// walks ONE sequence along the positions of its wave's stretch, so a run is seen from its first to its last position; the
// counting pass counts the runs, the encoding pass writes them (sequence << 32 | start, end) through the same counter.  A
// run that crosses the end of a stretch is listed in pieces — in both passes alike.
// ------------------------------------------------------------------------------------------------
struct MissingRunTracker {
   bool in_run = false;
   uint32_t start = 0;
};

/// Called by every lane of the wave: a run of `sequence` ends at `end` (exclusive) in the lanes where `ends_here`.
__device__ __forceinline__ void emitMissingRun(const SeqStoreDev& store, bool ends_here, uint32_t start, uint32_t end, uint64_t sequence) {
   const uint64_t ending = __ballot(ends_here);
   if (ending == 0) {
      return;
   }
   const uint32_t lane = threadIdx.x & 63u;
   const uint32_t leader = static_cast<uint32_t>(__builtin_ctzll(ending));
   unsigned long long first = 0;
   if (lane == leader) {
      first = atomicAdd(store.enc_run_count, static_cast<unsigned long long>(__popcll(ending)));
   }
   if (store.build_mode == BUILD_ENCODE) {
      first = __shfl(first, leader);
      const unsigned long long slot = first + static_cast<unsigned long long>(__popcll(ending & ((1ull << lane) - 1ull)));
      if (ends_here && slot < store.enc_run_capacity) {
         store.enc_run_keys[slot] = (sequence << 32) | start;
         store.enc_run_ends[slot] = end;
      }
   }
}

__device__ __forceinline__ void trackMissingRun(const SeqStoreDev& store, MissingRunTracker& tracker, uint32_t position, bool missing, uint64_t sequence) {
   if (store.runs_at_build == 0) {
      return;
   }
   emitMissingRun(store, tracker.in_run && !missing, tracker.start, position, sequence);
   if (missing && !tracker.in_run) {
      tracker.start = position;
   }
   tracker.in_run = missing;
}

__device__ __forceinline__ void flushMissingRun(const SeqStoreDev& store, MissingRunTracker& tracker, uint32_t end, uint64_t sequence) {
   if (store.runs_at_build != 0) {
      emitMissingRun(store, tracker.in_run, tracker.start, end, sequence);
      tracker.in_run = false;
   }
}

// ------------------------------------------------------------------------------------------------
// B1: transpose a batch of aligned sequences.  A wave owns one 64-sequence word and a range of
// positions; each lane reads 4 positions of its own sequence per load from the pitched staging
// buffer, so consecutive loads of a lane walk the same cache lines.
// ------------------------------------------------------------------------------------------------
constexpr uint32_t TRANSPOSE_POSITIONS_PER_WAVE = 256;

__global__ __launch_bounds__(256) void k_transpose_sequences(
   const SeqStoreDev store,
   const uint8_t* __restrict__ chars,  // [n][pitch]
   const uint8_t* __restrict__ is_null,
   uint32_t pitch,
   uint32_t first_sequence,
   uint32_t n_sequences,
   uint32_t first_word,
   uint32_t n_words,
   const uint8_t* __restrict__ char_table,  // [256]
   uint64_t* sparse,
   uint32_t* sparse_count,
   uint32_t sparse_capacity,
   uint32_t* error_flag
) {
   __shared__ uint8_t s_table[256];
   s_table[threadIdx.x] = char_table[threadIdx.x];
   __syncthreads();

   const uint32_t lane = threadIdx.x & 63u;
   const uint32_t wave_in_block = threadIdx.x >> 6;
   const uint32_t word_index = blockIdx.x * 4 + wave_in_block;
   if (word_index >= n_words) {
      return;
   }
   const uint32_t word = first_word + word_index;
   const uint64_t sequence = static_cast<uint64_t>(word) * 64u + lane;
   const bool active = sequence >= first_sequence && sequence < static_cast<uint64_t>(first_sequence) + n_sequences;
   const uint32_t local = active ? static_cast<uint32_t>(sequence - first_sequence) : 0;
   const bool null_genome = active && is_null != nullptr && is_null[local] != 0;
   // the word is overwritten only if all 64 of its sequences are in this batch
   const bool whole_word = static_cast<uint64_t>(word) * 64u >= first_sequence &&
                           static_cast<uint64_t>(word) * 64u + 64u <= static_cast<uint64_t>(first_sequence) + n_sequences;

   const uint32_t pos_begin = blockIdx.y * TRANSPOSE_POSITIONS_PER_WAVE;
   const uint32_t pos_end = min(store.positions, pos_begin + TRANSPOSE_POSITIONS_PER_WAVE);
   const uint8_t* row = chars + static_cast<size_t>(local) * pitch;
   MissingRunTracker missing_run;
   for (uint32_t p4 = pos_begin; p4 < pos_end; p4 += 4) {
      uint32_t packed = 0;
      if (active && !null_genome) {  // rows are contiguous (pitch = positions, any alignment): byte loads, served from L1
#pragma unroll
         for (uint32_t k = 0; k < 4; ++k) {
            if (p4 + k < pos_end) {
               packed |= static_cast<uint32_t>(row[p4 + k]) << (8 * k);
            }
         }
      }
      for (uint32_t k = 0; k < 4 && p4 + k < pos_end; ++k) {
         uint32_t symbol = SILO_GPU_SYMBOL_NONE;
         if (null_genome) {
            symbol = store.missing_symbol;
         } else if (active) {
            symbol = s_table[(packed >> (8 * k)) & 0xFFu];
            if (symbol == SILO_GPU_SYMBOL_NONE) {
               atomicOr(error_flag, 1u);
            }
         }
         trackMissingRun(store, missing_run, p4 + k, symbol == store.missing_symbol, sequence);
         emitWord(store, p4 + k, word, symbol, whole_word, sparse, sparse_count, sparse_capacity);
      }
   }
   flushMissingRun(store, missing_run, pos_end, sequence);
}

// ------------------------------------------------------------------------------------------------
// B2: synthetic planes (CPU twin: oracle/synth.py symbol_matrix()).
// ------------------------------------------------------------------------------------------------
__host__ __device__ inline uint64_t mix64(uint64_t z) {  // splitmix64 finaliser
   z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
   z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
   return z ^ (z >> 31);
}

struct SynthArgs {
   uint64_t seed;
   uint32_t n_lineages;
   uint32_t sequence_count;
   const uint16_t* lineage;
   const uint32_t* lead_gap;
   const uint32_t* trail_gap;
   const uint32_t* missing_start;
   const uint32_t* missing_len;
   const uint8_t* lineage_symbol;  // [P][L]
   const uint8_t* reference;       // [P]
   uint32_t private_threshold;
   uint32_t ambiguous_threshold;
   uint32_t private_base, private_count;      // nuc: 1,4 (A C G T)   aa: 1,20 (A..Y)
   uint32_t ambiguous_base, ambiguous_count;  // nuc: 5,10 (R..V)     aa: 21,2 (B Z)
   uint32_t position_offset;                  // global position of the store's position 0
   uint32_t total_positions;                  // genome length (>= position_offset + store positions)
};

constexpr uint32_t SYNTH_POSITIONS_PER_WAVE = 128;

__global__ __launch_bounds__(256) void k_generate_synthetic(
   const SeqStoreDev store, const SynthArgs args, uint32_t n_words, uint64_t* sparse, uint32_t* sparse_count,
   uint32_t sparse_capacity
) {
   const uint32_t lane = threadIdx.x & 63u;
   const uint32_t word = blockIdx.x * 4 + (threadIdx.x >> 6);
   if (word >= n_words) {
      return;
   }
   const uint64_t sequence = static_cast<uint64_t>(word) * 64u + lane;
   const bool active = sequence < args.sequence_count;
   const uint32_t i = active ? static_cast<uint32_t>(sequence) : 0;
   const uint32_t lineage = args.lineage[i];
   const uint32_t lead = args.lead_gap[i];
   const uint32_t trail = args.trail_gap[i];
   const uint32_t mstart = args.missing_start[i];
   const uint32_t mlen = args.missing_len[i];
   const uint32_t positions = args.total_positions;
   const uint64_t seq_hash = args.seed ^ (static_cast<uint64_t>(i) * 0x9E3779B97F4A7C15ull);

   const uint32_t pos_begin = blockIdx.y * SYNTH_POSITIONS_PER_WAVE;
   const uint32_t pos_end = min(store.positions, pos_begin + SYNTH_POSITIONS_PER_WAVE);
   MissingRunTracker missing_run;
   for (uint32_t local = pos_begin; local < pos_end; ++local) {
      const uint32_t p = args.position_offset + local;  // global genome position
      uint32_t symbol;
      if (p < lead || p >= positions - trail) {
         symbol = 0;  // GAP
      } else if (p >= mstart && p - mstart < mlen) {
         symbol = store.missing_symbol;
      } else {
         const uint64_t h = mix64(seq_hash ^ (static_cast<uint64_t>(p) * 0xC2B2AE3D27D4EB4Full));
         if ((h & 0xFFFFFu) < args.private_threshold) {
            symbol = args.private_base + static_cast<uint32_t>((h >> 20) & 0xFFFu) % args.private_count;
         } else if (((h >> 32) & 0xFFFFFFu) < args.ambiguous_threshold) {
            symbol = args.ambiguous_base + static_cast<uint32_t>(h >> 56) % args.ambiguous_count;
         } else {
            const uint8_t ls = args.lineage_symbol[static_cast<size_t>(local) * args.n_lineages + lineage];
            symbol = ls != SILO_GPU_SYMBOL_NONE ? ls : args.reference[local];
         }
      }
      if (!active) {
         symbol = SILO_GPU_SYMBOL_NONE;
      }
      trackMissingRun(store, missing_run, local, symbol == store.missing_symbol, sequence);
      emitWord(store, local, word, symbol, /*whole_word=*/true, sparse, sparse_count, sparse_capacity);
   }
   flushMissingRun(store, missing_run, pos_end, sequence);
}

__global__ __launch_bounds__(256) void k_bitset_from_lineages(
   const uint16_t* __restrict__ lineage, const uint8_t* __restrict__ membership, uint32_t sequence_count,
   uint32_t row_words, uint64_t* __restrict__ out
) {
   const uint32_t lane = threadIdx.x & 63u;
   const uint32_t word = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
   if (word >= row_words) {
      return;
   }
   const uint64_t sequence = static_cast<uint64_t>(word) * 64u + lane;
   const bool member = sequence < sequence_count && membership[lineage[sequence]] != 0;
   const uint64_t mask = __ballot(member);
   if (lane == 0) {
      out[word] = mask;
   }
}

__global__ __launch_bounds__(256) void k_bitset_from_value_ids(
   const uint32_t* __restrict__ value_ids, const uint8_t* __restrict__ membership, uint32_t n_values,
   uint32_t sequence_count, uint32_t row_words, uint64_t* __restrict__ out
) {
   const uint32_t lane = threadIdx.x & 63u;
   const uint32_t word = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
   if (word >= row_words) {
      return;
   }
   const uint64_t sequence = static_cast<uint64_t>(word) * 64u + lane;
   bool member = false;
   if (sequence < sequence_count) {
      const uint32_t value = value_ids[sequence];
      member = value < n_values && membership[value] != 0;
   }
   const uint64_t mask = __ballot(member);
   if (lane == 0) {
      out[word] = mask;
   }
}

__global__ void k_add_u32(uint32_t* __restrict__ dst, const uint32_t* __restrict__ src, uint32_t n) {
   const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
   if (i < n) {
      dst[i] += src[i];
   }
}

// K4: the row selection of Mutations::addMutationsToOutput (mutations.cpp:184-232) on the device: one thread per
// position sums its valid-symbol counts, applies the reference's threshold ceil(total * minProportion) - 1 in
// IEEE double exactly as the host code does, and appends the surviving (position, symbol) cells to a compact
// list.  The list is unordered (the host sorts a few hundred rows); past `capacity` only the counter advances.
__global__ __launch_bounds__(256) void k_mutations_select(
   const uint32_t* __restrict__ counts, const uint8_t* __restrict__ reference_index, uint32_t n_positions, uint32_t n_symbols,
   double min_proportion, uint32_t capacity, uint32_t* __restrict__ out
) {
   const uint32_t pos = blockIdx.x * blockDim.x + threadIdx.x;
   if (pos >= n_positions) {
      return;
   }
   const uint32_t* at_position = counts + static_cast<size_t>(pos) * n_symbols;
   uint32_t total = 0;
   for (uint32_t s = 0; s < n_symbols; ++s) {
      total += at_position[s];
   }
   if (total == 0) {
      return;
   }
   const uint32_t threshold_count =
      min_proportion == 0 ? 0u : static_cast<uint32_t>(ceil(static_cast<double>(total) * min_proportion) - 1);
   const uint32_t reference = reference_index[pos];
   uint32_t selected = 0;  // bit s: symbol s passes
   for (uint32_t s = 0; s < n_symbols; ++s) {
      if (s != reference && at_position[s] > threshold_count) {
         selected |= 1u << s;
      }
   }
   if (selected == 0) {
      return;
   }
   uint32_t slot = atomicAdd(&out[0], static_cast<uint32_t>(__popc(selected)));
   auto* rows = reinterpret_cast<silo_gpu_mutation_row*>(out + 4);
   for (uint32_t s = 0; s < n_symbols; ++s) {
      if ((selected >> s) & 1u) {
         if (slot < capacity) {
            rows[slot] = silo_gpu_mutation_row{pos, s, at_position[s], total};
         }
         ++slot;
      }
   }
}

// K4 with the list written straight into page-locked host memory (a row slot): no copy and no event between the scan and
// the host — the wait for a 6 KB device -> host copy and its event cost more than the row selection itself.  Rows go to the
// slot's host buffer (system-scope stores through the mapped pointer), the cursor and the ticket of finished blocks stay in
// device memory; every block makes its rows visible (system-scope fence) before it takes its ticket, and the block that
// takes the last one publishes epoch << 32 | number of selected cells (may exceed the capacity: then the caller falls back
// to the whole table) and re-arms cursor and ticket for the next launch.
__global__ __launch_bounds__(256) void k_mutations_select_to_host(
   const uint32_t* __restrict__ counts, const uint8_t* __restrict__ reference_index, uint32_t n_positions, uint32_t n_symbols,
   double min_proportion, uint32_t capacity, uint32_t* __restrict__ cursor_and_ticket, silo_gpu_mutation_row* __restrict__ host_rows,
   unsigned long long* __restrict__ host_header, uint32_t epoch
) {
   const uint32_t pos = blockIdx.x * blockDim.x + threadIdx.x;
   uint32_t selected = 0;  // bit s: symbol s passes
   uint32_t total = 0;
   const uint32_t* at_position = counts + static_cast<size_t>(pos) * n_symbols;
   if (pos < n_positions) {
      for (uint32_t s = 0; s < n_symbols; ++s) {
         total += at_position[s];
      }
      if (total != 0) {
         const uint32_t threshold_count = min_proportion == 0 ? 0u : static_cast<uint32_t>(ceil(static_cast<double>(total) * min_proportion) - 1);
         const uint32_t reference = reference_index[pos];
         for (uint32_t s = 0; s < n_symbols; ++s) {
            if (s != reference && at_position[s] > threshold_count) {
               selected |= 1u << s;
            }
         }
      }
   }
   if (selected != 0) {
      uint32_t slot = atomicAdd(&cursor_and_ticket[0], static_cast<uint32_t>(__popc(selected)));
      for (uint32_t s = 0; s < n_symbols; ++s) {
         if ((selected >> s) & 1u) {
            if (slot < capacity) {
               host_rows[slot] = silo_gpu_mutation_row{pos, s, at_position[s], total};
            }
            ++slot;
         }
      }
   }
   __threadfence_system();  // this thread's rows are in host memory ...
   __syncthreads();         // ... and so are those of the whole block, before its ticket is taken
   if (threadIdx.x == 0) {
      if (atomicAdd(&cursor_and_ticket[1], 1u) == gridDim.x - 1) {
         const uint32_t n_selected = atomicExch(&cursor_and_ticket[0], 0u);
         atomicExch(&cursor_and_ticket[1], 0u);
         __hip_atomic_store(host_header, (static_cast<unsigned long long>(epoch) << 32) | n_selected, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
      }
   }
}

__global__ void k_fill_ones(uint64_t* out, uint32_t row_words, uint32_t sequence_count) {
   const uint32_t w = blockIdx.x * blockDim.x + threadIdx.x;
   if (w < row_words) {
      out[w] = silo_gpu::valid_mask(w, sequence_count);
   }
}

// ------------------------------------------------------------------------------------------------
// The runs of the missing symbol.  k_missing_runs walks the plane [P][Wp] of the symbol along the positions: a wave owns one
// word column (64 consecutive sequences, one per lane), the 8 waves of a block the 8 columns of a 64-byte sector, so that the
// block reads every sector of the plane once.  WRITE = false counts the runs, WRITE = true emits them (sequence << 32 |
// start, end) through one atomic cursor; they are sorted afterwards.
// ------------------------------------------------------------------------------------------------
constexpr uint32_t RUN_BLOCK_THREADS = 512;

template <bool WRITE>
__global__ __launch_bounds__(RUN_BLOCK_THREADS) void k_missing_runs(
   const uint64_t* __restrict__ plane, uint32_t positions, uint32_t row_words, unsigned long long* __restrict__ n_runs, uint64_t* __restrict__ run_keys,
   uint32_t* __restrict__ run_ends, unsigned long long capacity
) {
   const uint32_t lane = threadIdx.x & 63u;
   const uint32_t word = blockIdx.x * (RUN_BLOCK_THREADS / 64) + (threadIdx.x >> 6);
   if (word >= row_words) {
      return;  // (uniform per wave)
   }
   const uint32_t sequence = word * 64u + lane;
   bool in_run = false;
   uint32_t start = 0;
   uint32_t counted = 0;
   const auto emit = [&](bool ends_here, uint32_t end) {
      if constexpr (WRITE) {
         const uint64_t ending = __ballot(ends_here);
         if (ending != 0) {
            unsigned long long first = 0;
            if (lane == static_cast<uint32_t>(__builtin_ctzll(ending))) {
               first = atomicAdd(n_runs, static_cast<unsigned long long>(__popcll(ending)));
            }
            first = __shfl(first, __builtin_ctzll(ending));
            const unsigned long long slot = first + static_cast<unsigned long long>(__popcll(ending & ((1ull << lane) - 1ull)));
            if (ends_here && slot < capacity) {
               run_keys[slot] = (static_cast<uint64_t>(sequence) << 32) | start;
               run_ends[slot] = end;
            }
         }
      } else {
         counted += ends_here ? 1u : 0u;
      }
   };
   constexpr uint32_t AHEAD = 8;  // plane words in flight per wave
   for (uint32_t base = 0; base < positions; base += AHEAD) {
      uint64_t words[AHEAD];
#pragma unroll
      for (uint32_t k = 0; k < AHEAD; ++k) {
         const uint32_t p = min(base + k, positions - 1u);
         words[k] = plane[static_cast<size_t>(p) * row_words + word];
      }
#pragma unroll
      for (uint32_t k = 0; k < AHEAD; ++k) {
         const uint32_t p = base + k;
         const bool inside = p < positions;  // (uniform) the last group may reach past the end: no state changes there
         const bool set = inside && ((words[k] >> lane) & 1ull) != 0;
         emit(inside && in_run && !set, p);
         if (set && !in_run) {
            start = p;
         }
         in_run = inside ? set : in_run;
      }
   }
   emit(in_run, positions);
   if constexpr (!WRITE) {
      const uint32_t wave_total = waveSumToLane63(counted);
      if (lane == 63u && wave_total != 0) {
         atomicAdd(n_runs, static_cast<unsigned long long>(wave_total));
      }
   }
}

/// The plane of the missing symbol at one position out of its runs: one thread per run (`out` zeroed beforehand).
__global__ __launch_bounds__(256) void k_runs_to_plane(
   const uint64_t* __restrict__ run_keys, const uint32_t* __restrict__ run_ends, uint32_t n_runs, uint32_t position, uint64_t* __restrict__ out
) {
   const uint32_t run = blockIdx.x * blockDim.x + threadIdx.x;
   if (run >= n_runs) {
      return;
   }
   const uint64_t key = run_keys[run];
   if (static_cast<uint32_t>(key) <= position && position < run_ends[run]) {
      const uint32_t sequence = static_cast<uint32_t>(key >> 32);
      atomicOr(reinterpret_cast<unsigned long long*>(out + (sequence >> 6)), 1ull << (sequence & 63u));
   }
}

/// Does `sequence` have the missing symbol at `position`?  The last run that starts at or before the cell decides.
__device__ __forceinline__ bool missingInRuns(const SeqStoreDev& store, uint32_t sequence, uint32_t position) {
   const uint64_t cell = (static_cast<uint64_t>(sequence) << 32) | position;
   uint32_t lo = 0, hi = store.n_missing_runs;
   while (lo < hi) {  // first run whose (sequence, start) is beyond the cell
      const uint32_t mid = lo + (hi - lo) / 2;
      if (store.missing_run_keys[mid] <= cell) {
         lo = mid + 1;
      } else {
         hi = mid;
      }
   }
   if (lo == 0) {
      return false;
   }
   const uint64_t key = store.missing_run_keys[lo - 1];
   return static_cast<uint32_t>(key >> 32) == sequence && position < store.missing_run_ends[lo - 1];
}

// FastaAligned: one thread per (requested row, position) looks the row's bit up in every dense plane of the
// position; a cell no dense plane claims holds a sparsely stored symbol (IUPAC code) and is found by binary search
// for position << 37 | symbol << 32 | sequence in the sorted sparse keys.  A gather (one 8-byte word per plane),
// sized for the <= 10 000 rows the action allows.
__global__ __launch_bounds__(256) void k_reconstruct_sequences(
   const SeqStoreDev store, const uint64_t* __restrict__ sparse_keys, uint32_t n_sparse, const uint32_t* __restrict__ row_ids,
   const char* __restrict__ symbol_chars, char* __restrict__ out
) {
   const uint32_t position = blockIdx.x * blockDim.x + threadIdx.x;
   if (position >= store.positions) {
      return;
   }
   const uint32_t sequence = row_ids[blockIdx.y];
   const uint32_t word = sequence >> 6;
   const uint32_t bit = sequence & 63u;
   uint32_t found = 0xFFu;
   // the row's code in the position's planes, and the valid mutation symbol that code stands for there (0 = none coded)
   const PositionLayout layout = layoutOf(store, position);
   const uint32_t code = codeOfRow(layout, store.row_words, word, bit);
   const uint32_t coded_index = code == 0 ? 0xFFu : (layout.identity ? code - 1u : layout.map[code]);
   for (uint32_t symbol = 0; symbol < store.n_symbols; ++symbol) {
      if (store.kind[symbol] == PLANE_SCAN) {
         if (store.index[symbol] == coded_index) {
            found = symbol;
         }
         continue;
      }
      if (store.kind[symbol] == PLANE_RUNS) {
         if (missingInRuns(store, sequence, position)) {
            found = symbol;
         }
         continue;
      }
      const uint64_t* plane = planePtr(store, position, symbol);
      if (plane != nullptr && ((plane[word] >> bit) & 1u) != 0) {
         found = symbol;
      }
   }
   const auto listed = [&](const uint64_t* keys, uint32_t lo, uint32_t hi, uint64_t key) {  // binary search in keys[lo, hi)
      const uint32_t end = hi;
      while (lo < hi) {
         const uint32_t mid = lo + (hi - lo) / 2;
         if (keys[mid] < key) {
            lo = mid + 1;
         } else {
            hi = mid;
         }
      }
      return lo < end && keys[lo] == key;
   };
   if (found == 0xFFu && store.escapes != nullptr) {  // a valid symbol that has no code at this position: an escape key
      const uint32_t first = store.escape_first[position];
      const uint32_t last = store.escape_first[position + 1];
      for (uint32_t symbol = 0; symbol < store.n_symbols && found == 0xFFu && first < last; ++symbol) {
         if (store.kind[symbol] == PLANE_SCAN &&
             listed(store.escapes, first, last, (static_cast<uint64_t>(position) << 37) | (static_cast<uint64_t>(store.index[symbol]) << 32) | sequence)) {
            found = symbol;
         }
      }
   }
   if (found == 0xFFu) {
      for (uint32_t symbol = 0; symbol < store.n_symbols && found == 0xFFu; ++symbol) {
         if (store.kind[symbol] == PLANE_SPARSE &&
             listed(sparse_keys, 0, n_sparse, (static_cast<uint64_t>(position) << 37) | (static_cast<uint64_t>(symbol) << 32) | sequence)) {
            found = symbol;
         }
      }
   }
   if (found == 0xFFu && layout.implicit) {  // no other symbol claims the cell: the position's derived symbol
      for (uint32_t symbol = 0; symbol < store.n_symbols; ++symbol) {
         if (store.kind[symbol] == PLANE_SCAN && store.index[symbol] == layout.map[IMPLICIT_SLOT]) {
            found = symbol;
         }
      }
   }
   out[static_cast<size_t>(blockIdx.y) * store.positions + position] = found == 0xFFu ? '?' : symbol_chars[found];
}

// One-hot plane of a valid mutation symbol out of the position's code planes (2, 3 or n_bits reads per word); a symbol
// that has no code at the position yields zeros (its rows are escape keys: the caller scatters them on top).
__global__ __launch_bounds__(256) void k_decode_plane(
   const SeqStoreDev store, uint32_t position, uint32_t symbol, const uint64_t* __restrict__ valid_words, uint64_t* __restrict__ out
) {
   const uint32_t word = blockIdx.x * blockDim.x + threadIdx.x;
   if (word < store.row_words) {
      const PositionLayout layout = layoutOf(store, position);
      const uint32_t code = codeOfSymbol(layout, store.index[symbol]);
      if (code == CODE_IMPLICIT) {  // the derived symbol: every row no stored row claims (the caller clears the keys, the runs, the sparse symbols)
         uint64_t others = 0;
         for (uint32_t row = 0; row < layout.bits; ++row) {
            others |= layout.rows[static_cast<size_t>(row) * store.row_words + word];
         }
         out[word] = ~others & valid_words[word];
         return;
      }
      out[word] = code == CODE_ESCAPED ? 0ull : decodeCodeWord(layout, store.row_words, code, word);
   }
}

/// Clears the rows of keys[begin, end) (sequence in the low 32 bits) in `out`.
__global__ void k_clear_keys(const uint64_t* __restrict__ keys, uint32_t begin, uint32_t end, uint64_t* out) {
   const uint32_t k = begin + blockIdx.x * blockDim.x + threadIdx.x;
   if (k < end) {
      const uint32_t sequence = static_cast<uint32_t>(keys[k] & 0xFFFFFFFFull);
      atomicAnd(reinterpret_cast<unsigned long long*>(out + (sequence >> 6)), ~(1ull << (sequence & 63u)));
   }
}

/// Clears the rows whose run of the missing symbol covers `position` in `out`.
__global__ __launch_bounds__(256) void k_runs_clear_plane(
   const uint64_t* __restrict__ run_keys, const uint32_t* __restrict__ run_ends, uint32_t n_runs, uint32_t position, uint64_t* __restrict__ out
) {
   const uint32_t run = blockIdx.x * blockDim.x + threadIdx.x;
   if (run >= n_runs) {
      return;
   }
   const uint64_t key = run_keys[run];
   if (static_cast<uint32_t>(key) <= position && position < run_ends[run]) {
      const uint32_t sequence = static_cast<uint32_t>(key >> 32);
      atomicAnd(reinterpret_cast<unsigned long long*>(out + (sequence >> 6)), ~(1ull << (sequence & 63u)));
   }
}

__global__ void k_scatter_sparse(const uint64_t* __restrict__ keys, uint32_t begin, uint32_t end, uint64_t* out) {
   const uint32_t k = begin + blockIdx.x * blockDim.x + threadIdx.x;
   if (k < end) {
      const uint32_t sequence = static_cast<uint32_t>(keys[k] & 0xFFFFFFFFull);
      atomicOr(reinterpret_cast<unsigned long long*>(out + (sequence >> 6)), 1ull << (sequence & 63u));
   }
}

// ------------------------------------------------------------------------------------------------
// host helpers
// ------------------------------------------------------------------------------------------------
int ensureDevice(int device) {
   int count = 0;
   hipError_t err = hipGetDeviceCount(&count);
   if (err != hipSuccess || count == 0) {
      return fail(
         SILO_GPU_ERR_NO_DEVICE,
         "no HIP device visible (hipGetDeviceCount: " + std::string(hipGetErrorString(err)) +
            "); the silo_gpu product path has no CPU fallback"
      );
   }
   if (device < 0 || device >= count) {
      return fail(SILO_GPU_ERR_INVALID_ARGUMENT, "device ordinal out of range");
   }
   HIP_TRY(hipSetDevice(device));
   return SILO_GPU_OK;
}

int growSparse(SeqStoreHost& seqstore, uint32_t needed) {
   if (needed <= seqstore.sparse_capacity) {
      return SILO_GPU_OK;
   }
   uint32_t capacity = std::max<uint32_t>(1u << 16, seqstore.sparse_capacity);
   while (capacity < needed) {
      capacity *= 2;
   }
   uint64_t* bigger = nullptr;
   HIP_TRY(hipMalloc(&bigger, static_cast<size_t>(capacity) * sizeof(uint64_t)));
   if (seqstore.d_sparse != nullptr) {
      HIP_TRY(hipMemcpy(bigger, seqstore.d_sparse, static_cast<size_t>(seqstore.sparse_capacity) * sizeof(uint64_t), hipMemcpyDeviceToDevice));
      HIP_TRY(hipFree(seqstore.d_sparse));
   }
   seqstore.d_sparse = bigger;
   seqstore.sparse_capacity = capacity;
   return SILO_GPU_OK;
}

/// The build-time planes of a sequence store, allocated (zeroed) when its first sequences arrive.
int ensureBuildPlanes(silo_gpu_store* store, SeqStoreHost& seqstore) {
   if (seqstore.layout.built) {
      return fail(
         SILO_GPU_ERR_INVALID_ARGUMENT, "the sequence store is finalized: its build-time planes were re-encoded and released, no sequences can be added"
      );
   }
   SeqStoreDev& dev = seqstore.dev;
   if (dev.scan != nullptr || dev.extra != nullptr || dev.build_mode == BUILD_COUNT) {
      return SILO_GPU_OK;  // (the counting pass of a two-pass build writes no plane at all)
   }
   // the encoding pass of a two-pass build writes the valid symbols straight into the adaptive planes: only the extra planes are built
   const size_t scan_bytes = dev.build_mode == BUILD_ENCODE ? 0 : static_cast<size_t>(dev.positions) * dev.n_bits * dev.row_words * sizeof(uint64_t);
   const size_t extra_bytes = dev.runs_at_build != 0 ? 0 : static_cast<size_t>(dev.positions) * dev.n_extra * dev.row_words * sizeof(uint64_t);
   if (scan_bytes > 0) {
      HIP_TRY(hipMalloc(&dev.scan, scan_bytes));
      HIP_TRY(hipMemset(dev.scan, 0, scan_bytes));
   }
   if (extra_bytes > 0) {
      HIP_TRY(hipMalloc(&dev.extra, extra_bytes));
      HIP_TRY(hipMemset(dev.extra, 0, extra_bytes));
   }
   dev.planes = dev.scan;
   store->device_bytes += scan_bytes + extra_bytes;
   return SILO_GPU_OK;
}

}  // namespace

// ================================================================================================
// C ABI
// ================================================================================================
extern "C" {

const char* silo_gpu_last_error(void) {
   return g_last_error.c_str();
}

const char* silo_gpu_last_scan_kernel(void) {
   return g_last_scan_kernel;
}

int silo_gpu_tune(int knob, int value) {
   if (knob == SILO_GPU_TUNE_SCAN_ROWS_PER_BLOCK) {
      return g_tune_rows_per_block.exchange(value);
   }
   if (knob == SILO_GPU_TUNE_SCAN_VARIANT) {
      return g_tune_scan_variant.exchange(value);
   }
   if (knob == SILO_GPU_TUNE_EVAL_LEAF_BATCH) {
      return g_tune_eval_leaf_batch.exchange(value);
   }
   if (knob == SILO_GPU_TUNE_SCAN_SPARSE_DIVISOR) {
      return g_tune_sparse_divisor.exchange(value);
   }
   if (knob == SILO_GPU_TUNE_COMPACT_INDEX) {
      return g_tune_compact_index.exchange(value);
   }
   if (knob == SILO_GPU_TUNE_SIDE_STREAM) {
      return g_tune_side_stream.exchange(value);
   }
   if (knob == SILO_GPU_TUNE_KEY_COST) {
      return g_tune_key_cost.exchange(value);
   }
   if (knob == SILO_GPU_TUNE_SCAN_TIMING) {
      return g_tune_scan_timing.exchange(value);
   }
   if (knob == SILO_GPU_TUNE_MISSING_RUNS) {
      return g_tune_missing_runs.exchange(value);
   }
   if (knob == SILO_GPU_TUNE_LAUNCH_COST) {
      return g_tune_launch_cost.exchange(value);
   }
   return -1;
}

int silo_gpu_store_create(const silo_gpu_store_desc* desc, silo_gpu_store** out) {
   if (desc == nullptr || out == nullptr || desc->n_seqstores == 0 || desc->seqstores == nullptr) {
      return fail(SILO_GPU_ERR_INVALID_ARGUMENT, "silo_gpu_store_create: null descriptor");
   }
   *out = nullptr;
   if (int rc = ensureDevice(desc->device); rc != SILO_GPU_OK) {
      return rc;
   }
   auto* store = new (std::nothrow) silo_gpu_store();
   if (store == nullptr) {
      return fail(SILO_GPU_ERR_OUT_OF_MEMORY, "host allocation failed");
   }
   store->device = desc->device;
   store->sequence_count = desc->sequence_count;
   const uint32_t words = (desc->sequence_count + 63u) / 64u;
   store->row_words = std::max(ROW_ALIGN_WORDS, (words + ROW_ALIGN_WORDS - 1) / ROW_ALIGN_WORDS * ROW_ALIGN_WORDS);
   const uint32_t row_words = store->row_words;

   auto cleanup = [&](int code) {
      silo_gpu_store_destroy(store);
      return code;
   };

   store->seqstores.resize(desc->n_seqstores);
   for (uint32_t k = 0; k < desc->n_seqstores; ++k) {
      const silo_gpu_seqstore_desc& in = desc->seqstores[k];
      SeqStoreHost& seqstore = store->seqstores[k];
      if (in.alphabet > SILO_GPU_ALPHABET_AMINO_ACID || in.positions == 0 || in.reference == nullptr) {
         return cleanup(fail(SILO_GPU_ERR_INVALID_ARGUMENT, "invalid sequence store descriptor"));
      }
      seqstore.alphabet = in.alphabet;
      seqstore.reference.assign(in.reference, in.reference + in.positions);
      SeqStoreDev& dev = seqstore.dev;
      dev.positions = in.positions;
      dev.n_symbols = alphabetSize(in.alphabet);
      dev.n_scan = in.n_scan_symbols;
      dev.n_bits = 0;
      while ((1u << dev.n_bits) < dev.n_scan + 1u) {
         ++dev.n_bits;
      }
      dev.n_extra = in.n_extra_symbols;
      dev.row_words = row_words;
      dev.missing_symbol = missingSymbol(in.alphabet);
      for (uint32_t s = 0; s < SILO_GPU_MAX_SYMBOLS; ++s) {
         dev.kind[s] = PLANE_SPARSE;
         dev.index[s] = 0;
      }
      for (uint32_t s = 0; s < in.n_scan_symbols; ++s) {
         if (in.scan_symbols[s] >= dev.n_symbols) {
            return cleanup(fail(SILO_GPU_ERR_INVALID_ARGUMENT, "scan symbol out of range"));
         }
         dev.kind[in.scan_symbols[s]] = PLANE_SCAN;
         dev.index[in.scan_symbols[s]] = static_cast<uint8_t>(s);
      }
      for (uint32_t s = 0; s < in.n_extra_symbols; ++s) {
         if (in.extra_symbols[s] >= dev.n_symbols || dev.kind[in.extra_symbols[s]] != PLANE_SPARSE) {
            return cleanup(fail(SILO_GPU_ERR_INVALID_ARGUMENT, "extra symbol out of range or duplicated"));
         }
         dev.kind[in.extra_symbols[s]] = PLANE_EXTRA;
         dev.index[in.extra_symbols[s]] = static_cast<uint8_t>(s);
      }
      // the planes are allocated when the first sequences arrive (ensureBuildPlanes) and re-encoded at finalize: stores
      // that are filled and finalized one after the other never hold their build-time planes at the same time
      hipError_t err = hipSuccess;
      if (err == hipSuccess) {
         err = hipMalloc(&seqstore.d_reference, in.positions);
      }
      if (err == hipSuccess) {
         err = hipMemcpy(seqstore.d_reference, in.reference, in.positions, hipMemcpyHostToDevice);
      }
      if (err == hipSuccess) {
         err = hipMalloc(&seqstore.d_sparse_count, sizeof(uint32_t));
      }
      if (err == hipSuccess) {
         err = hipMemset(seqstore.d_sparse_count, 0, sizeof(uint32_t));
      }
      if (err != hipSuccess) {
         return cleanup(fail(
            err == hipErrorOutOfMemory ? SILO_GPU_ERR_OUT_OF_MEMORY : SILO_GPU_ERR_HIP,
            std::string("allocating planes: ") + hipGetErrorString(err)
         ));
      }
   }
   hipError_t err = hipMalloc(&store->d_ones, static_cast<size_t>(row_words) * sizeof(uint64_t));
   if (err == hipSuccess) {
      err = hipMalloc(&store->d_error_flag, sizeof(uint32_t));
   }
   if (err == hipSuccess) {
      err = hipMemset(store->d_error_flag, 0, sizeof(uint32_t));
   }
   if (err != hipSuccess) {
      return cleanup(fail(SILO_GPU_ERR_HIP, std::string("allocating store: ") + hipGetErrorString(err)));
   }
   k_fill_ones<<<(row_words + 255) / 256, 256>>>(store->d_ones, row_words, store->sequence_count);
   err = hipDeviceSynchronize();
   if (err != hipSuccess) {
      return cleanup(fail(SILO_GPU_ERR_HIP, std::string("k_fill_ones: ") + hipGetErrorString(err)));
   }
   *out = store;
   return SILO_GPU_OK;
}

void silo_gpu_store_destroy(silo_gpu_store* store) {
   if (store == nullptr) {
      return;
   }
   (void)hipSetDevice(store->device);
   for (SeqStoreHost& seqstore : store->seqstores) {
      (void)hipFree(seqstore.dev.scan);
      (void)hipFree(seqstore.dev.extra);
      (void)hipFree(seqstore.d_reference);
      (void)hipFree(seqstore.d_sparse);
      (void)hipFree(seqstore.d_sparse_count);
      (void)hipFree(seqstore.d_totals);
      (void)hipFree(seqstore.d_missing_run_keys);
      (void)hipFree(seqstore.d_missing_run_ends);
      if (seqstore.work) {  // a two-pass build that was never finalized
         seqstore.work->discard();
      }
      (void)hipFree(seqstore.d_run_count);
      (void)hipFree(seqstore.layout.planes);
      (void)hipFree(seqstore.layout.d_row_of);
      (void)hipFree(seqstore.layout.d_row_target);
      (void)hipFree(seqstore.layout.d_code_map);
      (void)hipFree(seqstore.layout.d_escapes);
      (void)hipFree(seqstore.layout.d_escapes_sliced);
      (void)hipFree(seqstore.layout.d_slice_first);
      (void)hipFree(seqstore.layout.d_run_slice_first);
      (void)hipFree(seqstore.layout.d_escape_first);
   }
   (void)hipFree(store->d_ones);
   (void)hipFree(store->d_lineage);
   (void)hipFree(store->d_error_flag);
   (void)hipFree(store->d_stage);
   (void)hipFree(store->d_stage_null);
   (void)hipFree(store->d_import_row);
   (void)hipFree(store->d_import_union);
   (void)hipFree(store->d_char_table[0]);
   (void)hipFree(store->d_char_table[1]);
   (void)hipFree(store->d_symbol_chars[0]);
   (void)hipFree(store->d_symbol_chars[1]);
   delete store;
}

int silo_gpu_store_set_options(silo_gpu_store* store, const silo_gpu_store_options* options) {
   if (store == nullptr) {
      return fail(SILO_GPU_ERR_INVALID_ARGUMENT, "silo_gpu_store_set_options: null store");
   }
   std::lock_guard<std::mutex> lock(store->mutex);
   store->options = options != nullptr ? *options
                                       : silo_gpu_store_options{SILO_GPU_OPTION_DEFAULT, SILO_GPU_OPTION_DEFAULT, SILO_GPU_OPTION_DEFAULT, SILO_GPU_OPTION_DEFAULT};
   return SILO_GPU_OK;
}

uint32_t silo_gpu_store_sequence_count(const silo_gpu_store* store) {
   return store != nullptr ? store->sequence_count : 0;
}
uint32_t silo_gpu_store_row_words(const silo_gpu_store* store) {
   return store != nullptr ? store->row_words : 0;
}
int silo_gpu_store_memory_info(const silo_gpu_store* store, uint64_t* free_bytes, uint64_t* total_bytes) {
   if (store == nullptr || free_bytes == nullptr || total_bytes == nullptr) {
      return fail(SILO_GPU_ERR_INVALID_ARGUMENT, "silo_gpu_store_memory_info: null argument");
   }
   HIP_TRY(hipSetDevice(store->device));
   size_t free_now = 0, total = 0;
   HIP_TRY(hipMemGetInfo(&free_now, &total));
   *free_bytes = free_now;
   *total_bytes = total;
   return SILO_GPU_OK;
}

uint64_t silo_gpu_store_device_bytes(const silo_gpu_store* store) {
   return store != nullptr ? store->device_bytes : 0;
}

int silo_gpu_store_append_sequences(
   silo_gpu_store* store, uint32_t seqstore_id, uint32_t first_sequence, uint32_t n_sequences, const char* chars,
   const uint8_t* is_null
) {
   if (store == nullptr || seqstore_id >= store->seqstores.size() || (chars == nullptr && n_sequences > 0)) {
      return fail(SILO_GPU_ERR_INVALID_ARGUMENT, "silo_gpu_store_append_sequences: bad arguments");
   }
   if (static_cast<uint64_t>(first_sequence) + n_sequences > store->sequence_count) {
      return fail(SILO_GPU_ERR_INVALID_ARGUMENT, "append beyond sequence_count");
   }
   if (n_sequences == 0) {
      return SILO_GPU_OK;
   }
   std::lock_guard<std::mutex> lock(store->mutex);
   HIP_TRY(hipSetDevice(store->device));
   SeqStoreHost& seqstore = store->seqstores[seqstore_id];
   if (const int rc = ensureBuildPlanes(store, seqstore); rc != SILO_GPU_OK) {
      return rc;
   }
   seqstore.finalized = false;
   seqstore.totals_ready = seqstore.dev.build_mode == BUILD_ENCODE;  // (the counts of the first pass ARE the totals)
   if (seqstore.dev.build_mode != BUILD_ENCODE) {
      seqstore.rows_filled += n_sequences;
   }
   const uint32_t positions = seqstore.dev.positions;
   const uint32_t pitch = positions;  // rows stay contiguous: ONE host-to-device copy per batch

   const size_t stage_bytes = static_cast<size_t>(n_sequences) * pitch;
   if (stage_bytes > store->stage_capacity) {
      (void)hipFree(store->d_stage);
      store->d_stage = nullptr;
      store->stage_capacity = 0;
      HIP_TRY(hipMalloc(&store->d_stage, stage_bytes));
      store->stage_capacity = stage_bytes;
   }
   if (is_null != nullptr && n_sequences > store->stage_null_capacity) {
      (void)hipFree(store->d_stage_null);
      store->d_stage_null = nullptr;
      store->stage_null_capacity = 0;
      HIP_TRY(hipMalloc(&store->d_stage_null, n_sequences));
      store->stage_null_capacity = n_sequences;
   }
   uint8_t*& d_table_slot = store->d_char_table[seqstore.alphabet == SILO_GPU_ALPHABET_NUCLEOTIDE ? 0 : 1];
   if (d_table_slot == nullptr) {
      uint8_t table[256];
      fillCharTable(seqstore.alphabet, table);
      HIP_TRY(hipMalloc(&d_table_slot, 256));
      HIP_TRY(hipMemcpy(d_table_slot, table, 256, hipMemcpyHostToDevice));
   }
   uint8_t* d_chars = store->d_stage;
   uint8_t* d_null = is_null != nullptr ? store->d_stage_null : nullptr;
   uint8_t* d_table = d_table_slot;
   auto release = [] {};  // staging is owned by the store
   HIP_TRY(hipMemcpy(d_chars, chars, stage_bytes, hipMemcpyHostToDevice));
   if (is_null != nullptr) {
      HIP_TRY(hipMemcpy(d_null, is_null, n_sequences, hipMemcpyHostToDevice));
   }
   hipError_t err = hipSuccess;

   const uint32_t first_word = first_sequence / 64u;
   const uint32_t last_word = (first_sequence + n_sequences - 1u) / 64u;
   const uint32_t n_words = last_word - first_word + 1u;
   const dim3 grid((n_words + 3) / 4, (positions + TRANSPOSE_POSITIONS_PER_WAVE - 1) / TRANSPOSE_POSITIONS_PER_WAVE);

   // The passes of a two-pass build must not be replayed — the counting pass adds to counters, the encoding pass takes key
   // slots and run slots with atomic cursors — so they never overflow the sparse buffer: the counting pass stores no sparse key
   // at all (capacity 0: the counter counts them), silo_gpu_store_build_pass(2) sizes the buffer from that count.
   const uint32_t build_mode = seqstore.dev.build_mode;
   uint32_t count_before = 0;
   err = hipMemcpy(&count_before, seqstore.d_sparse_count, sizeof(uint32_t), hipMemcpyDeviceToHost);
   if (err != hipSuccess) {
      release();
      return fail(SILO_GPU_ERR_HIP, std::string("reading sparse counter: ") + hipGetErrorString(err));
   }
   if (build_mode == BUILD_PLANES) {
      if (int rc = growSparse(seqstore, count_before + (1u << 16)); rc != SILO_GPU_OK) {
         release();
         return rc;
      }
   }
   // The dense writes of an ordinary build are idempotent (atomicOr / whole-word stores); if the sparse buffer overflows
   // the counter is rewound, the buffer grown and the batch replayed.
   for (int attempt = 0; attempt < 8; ++attempt) {
      k_transpose_sequences<<<grid, 256>>>(
         seqstore.dev, d_chars, d_null, pitch, first_sequence, n_sequences, first_word, n_words, d_table,
         seqstore.d_sparse, seqstore.d_sparse_count, build_mode == BUILD_COUNT ? 0u : seqstore.sparse_capacity, store->d_error_flag
      );
      err = hipDeviceSynchronize();
      uint32_t count_after = 0;
      if (err == hipSuccess) {
         err = hipMemcpy(&count_after, seqstore.d_sparse_count, sizeof(uint32_t), hipMemcpyDeviceToHost);
      }
      if (err != hipSuccess) {
         release();
         return fail(SILO_GPU_ERR_HIP, std::string("k_transpose_sequences: ") + hipGetErrorString(err));
      }
      if (build_mode == BUILD_COUNT || count_after <= seqstore.sparse_capacity) {
         break;
      }
      if (build_mode == BUILD_ENCODE) {
         release();
         return fail(SILO_GPU_ERR_INVALID_ARGUMENT, "two-pass build: the second pass brought more sparsely stored symbols than the first pass counted");
      }
      err = hipMemcpy(seqstore.d_sparse_count, &count_before, sizeof(uint32_t), hipMemcpyHostToDevice);
      if (err != hipSuccess) {
         release();
         return fail(SILO_GPU_ERR_HIP, std::string("rewinding sparse counter: ") + hipGetErrorString(err));
      }
      if (int rc = growSparse(seqstore, count_after); rc != SILO_GPU_OK) {
         release();
         return rc;
      }
   }
   release();
   uint32_t error_flag = 0;
   HIP_TRY(hipMemcpy(&error_flag, store->d_error_flag, sizeof(uint32_t), hipMemcpyDeviceToHost));
   if (error_flag != 0) {
      HIP_TRY(hipMemset(store->d_error_flag, 0, sizeof(uint32_t)));
      return fail(SILO_GPU_ERR_INVALID_ARGUMENT, "Illegal character contained in sequence.");
   }
   return SILO_GPU_OK;
}

int silo_gpu_store_generate_synthetic(silo_gpu_store* store, uint32_t seqstore_id, const silo_gpu_synth_desc* synth) {
   if (store == nullptr || synth == nullptr || seqstore_id >= store->seqstores.size() || synth->n_lineages == 0) {
      return fail(SILO_GPU_ERR_INVALID_ARGUMENT, "silo_gpu_store_generate_synthetic: bad arguments");
   }
   std::lock_guard<std::mutex> lock(store->mutex);
   HIP_TRY(hipSetDevice(store->device));
   SeqStoreHost& seqstore = store->seqstores[seqstore_id];
   if (const int rc = ensureBuildPlanes(store, seqstore); rc != SILO_GPU_OK) {
      return rc;
   }
   seqstore.finalized = false;
   seqstore.totals_ready = seqstore.dev.build_mode == BUILD_ENCODE;
   seqstore.rows_filled = store->sequence_count;
   const uint32_t n = store->sequence_count;
   const uint32_t positions = seqstore.dev.positions;

   uint32_t* d_u32[4] = {nullptr, nullptr, nullptr, nullptr};
   uint8_t* d_lineage_symbol = nullptr;
   auto release = [&]() {
      for (auto* ptr : d_u32) {
         (void)hipFree(ptr);
      }
      (void)hipFree(d_lineage_symbol);
   };
   hipError_t err = hipSuccess;
   if (store->d_lineage == nullptr) {
      err = hipMalloc(&store->d_lineage, static_cast<size_t>(n) * sizeof(uint16_t));
   }
   if (err == hipSuccess) {
      err = hipMemcpy(store->d_lineage, synth->lineage_of_sequence, static_cast<size_t>(n) * sizeof(uint16_t), hipMemcpyHostToDevice);
      store->n_lineages = synth->n_lineages;
   }
   const uint32_t* host_u32[4] = {synth->lead_gap, synth->trail_gap, synth->missing_start, synth->missing_len};
   for (int k = 0; k < 4 && err == hipSuccess; ++k) {
      err = hipMalloc(&d_u32[k], static_cast<size_t>(n) * sizeof(uint32_t));
      if (err == hipSuccess) {
         err = hipMemcpy(d_u32[k], host_u32[k], static_cast<size_t>(n) * sizeof(uint32_t), hipMemcpyHostToDevice);
      }
   }
   const size_t table_bytes = static_cast<size_t>(positions) * synth->n_lineages;
   if (err == hipSuccess) {
      err = hipMalloc(&d_lineage_symbol, table_bytes);
   }
   if (err == hipSuccess) {
      err = hipMemcpy(d_lineage_symbol, synth->lineage_symbol, table_bytes, hipMemcpyHostToDevice);
   }
   if (err != hipSuccess) {
      release();
      return fail(SILO_GPU_ERR_HIP, std::string("staging synthetic model: ") + hipGetErrorString(err));
   }

   SynthArgs args{};
   args.seed = synth->seed;
   args.n_lineages = synth->n_lineages;
   args.sequence_count = n;
   args.lineage = store->d_lineage;
   args.lead_gap = d_u32[0];
   args.trail_gap = d_u32[1];
   args.missing_start = d_u32[2];
   args.missing_len = d_u32[3];
   args.lineage_symbol = d_lineage_symbol;
   args.reference = seqstore.d_reference;
   args.private_threshold = synth->private_threshold;
   args.ambiguous_threshold = synth->ambiguous_threshold;
   args.position_offset = synth->position_offset;
   args.total_positions = synth->total_positions != 0 ? synth->total_positions : positions;
   if (static_cast<uint64_t>(args.position_offset) + positions > args.total_positions) {
      release();
      return fail(SILO_GPU_ERR_INVALID_ARGUMENT, "synthetic position window exceeds total_positions");
   }
   if (seqstore.alphabet == SILO_GPU_ALPHABET_NUCLEOTIDE) {
      args.private_base = 1;
      args.private_count = 4;
      args.ambiguous_base = 5;
      args.ambiguous_count = 10;
   } else {
      args.private_base = 1;
      args.private_count = 20;
      args.ambiguous_base = 21;
      args.ambiguous_count = 2;
   }

   const uint32_t n_words = (n + 63u) / 64u;
   const dim3 grid((n_words + 3) / 4, (positions + SYNTH_POSITIONS_PER_WAVE - 1) / SYNTH_POSITIONS_PER_WAVE);
   // expected sparse entries: cells * ambiguous_threshold / 2^24 (+ slack)
   const double expected = static_cast<double>(n) * positions * (static_cast<double>(synth->ambiguous_threshold) / 16777216.0);
   uint32_t zero = 0;
   err = hipMemcpy(seqstore.d_sparse_count, &zero, sizeof(uint32_t), hipMemcpyHostToDevice);
   if (err != hipSuccess) {
      release();
      return fail(SILO_GPU_ERR_HIP, std::string("resetting sparse counter: ") + hipGetErrorString(err));
   }
   const double wanted = expected * 1.25 + 65536.0;
   if (wanted > 4.0e9) {
      release();
      return fail(SILO_GPU_ERR_INVALID_ARGUMENT, "ambiguous_threshold too large for the sparse store");
   }
   const uint32_t build_mode = seqstore.dev.build_mode;  // (the passes of a two-pass build are never replayed: see append_sequences)
   if (build_mode == BUILD_PLANES) {
      if (int rc = growSparse(seqstore, static_cast<uint32_t>(wanted)); rc != SILO_GPU_OK) {
         release();
         return rc;
      }
   }
   for (int attempt = 0; attempt < 4; ++attempt) {
      k_generate_synthetic<<<grid, 256>>>(
         seqstore.dev, args, n_words, seqstore.d_sparse, seqstore.d_sparse_count, build_mode == BUILD_COUNT ? 0u : seqstore.sparse_capacity
      );
      err = hipDeviceSynchronize();
      uint32_t count_after = 0;
      if (err == hipSuccess) {
         err = hipMemcpy(&count_after, seqstore.d_sparse_count, sizeof(uint32_t), hipMemcpyDeviceToHost);
      }
      if (err != hipSuccess) {
         release();
         return fail(SILO_GPU_ERR_HIP, std::string("k_generate_synthetic: ") + hipGetErrorString(err));
      }
      if (build_mode == BUILD_COUNT || count_after <= seqstore.sparse_capacity) {
         break;
      }
      if (build_mode == BUILD_ENCODE) {
         release();
         return fail(SILO_GPU_ERR_INVALID_ARGUMENT, "two-pass build: the second pass brought more sparsely stored symbols than the first pass counted");
      }
      err = hipMemcpy(seqstore.d_sparse_count, &zero, sizeof(uint32_t), hipMemcpyHostToDevice);
      if (err != hipSuccess) {
         release();
         return fail(SILO_GPU_ERR_HIP, std::string("rewinding sparse counter: ") + hipGetErrorString(err));
      }
      if (int rc = growSparse(seqstore, count_after); rc != SILO_GPU_OK) {
         release();
         return rc;
      }
   }
   release();
   return SILO_GPU_OK;
}

namespace {
/// Sorts the sparse keys of one sequence store and re-encodes its build-time planes into the adaptive code planes.
/// finalize(): the plane of the missing symbol becomes the list of its runs (PLANE_RUNS) where that takes less than a quarter
/// of the plane — always, for data whose missing cells come in runs — and the plane is released.
int compactMissingPlane(silo_gpu_store* store, SeqStoreHost& seqstore) {
   SeqStoreDev& dev = seqstore.dev;
   if (dev.runs_at_build != 0) {  // a two-pass build wrote the runs while the rows streamed in: they only have to be put in order
      dev.runs_at_build = 0;
      unsigned long long written = 0;
      HIP_TRY(hipMemcpy(&written, seqstore.d_run_count, sizeof(written), hipMemcpyDeviceToHost));
      if (written != dev.enc_run_capacity) {
         return fail(SILO_GPU_ERR_INVALID_ARGUMENT, "two-pass build: the second pass did not bring the rows the first pass counted (runs of the missing symbol differ)");
      }
      if (const int rc = silo_gpu_internal_sort_pairs(seqstore.d_missing_run_keys, seqstore.d_missing_run_ends, written); rc != SILO_GPU_OK) {
         return rc;
      }
      store->device_bytes += std::max<size_t>(written, 1) * (sizeof(uint64_t) + sizeof(uint32_t));
      dev.missing_run_keys = seqstore.d_missing_run_keys;
      dev.missing_run_ends = seqstore.d_missing_run_ends;
      dev.n_missing_runs = static_cast<uint32_t>(written);
      dev.kind[dev.missing_symbol] = PLANE_RUNS;
      return SILO_GPU_OK;
   }
   if (dev.n_extra != 1 || dev.extra == nullptr || dev.kind[dev.missing_symbol] != PLANE_EXTRA || dev.positions == 0 || missingRunsOption(store) < 0) {
      return SILO_GPU_OK;
   }
   const size_t plane_bytes = static_cast<size_t>(dev.positions) * dev.row_words * sizeof(uint64_t);
   unsigned long long* d_count = nullptr;
   HIP_TRY(hipMalloc(&d_count, sizeof(unsigned long long)));
   const auto count_runs = [&](bool write, uint64_t* keys, uint32_t* ends, unsigned long long capacity, unsigned long long* out) -> int {
      HIP_TRY(hipMemset(d_count, 0, sizeof(unsigned long long)));
      const uint32_t blocks = (dev.row_words + RUN_BLOCK_THREADS / 64 - 1) / (RUN_BLOCK_THREADS / 64);
      if (write) {
         k_missing_runs<true><<<blocks, RUN_BLOCK_THREADS>>>(dev.extra, dev.positions, dev.row_words, d_count, keys, ends, capacity);
      } else {
         k_missing_runs<false><<<blocks, RUN_BLOCK_THREADS>>>(dev.extra, dev.positions, dev.row_words, d_count, keys, ends, capacity);
      }
      HIP_TRY(hipGetLastError());
      HIP_TRY(hipMemcpy(out, d_count, sizeof(unsigned long long), hipMemcpyDeviceToHost));
      return SILO_GPU_OK;
   };
   unsigned long long n_runs = 0;
   int rc = count_runs(false, nullptr, nullptr, 0, &n_runs);
   if (rc != SILO_GPU_OK || n_runs >= (1ull << 32) || n_runs * (sizeof(uint64_t) + sizeof(uint32_t)) > plane_bytes / 4) {
      (void)hipFree(d_count);
      return rc;  // scattered missing cells: the plane stays
   }
   uint64_t* d_keys = nullptr;
   uint32_t* d_ends = nullptr;
   const size_t slots = std::max<size_t>(n_runs, 1);
   hipError_t status = hipMalloc(&d_keys, slots * sizeof(uint64_t));
   status = status != hipSuccess ? status : hipMalloc(&d_ends, slots * sizeof(uint32_t));
   if (status == hipSuccess) {
      unsigned long long written = 0;
      rc = count_runs(true, d_keys, d_ends, n_runs, &written);
      if (rc == SILO_GPU_OK && written != n_runs) {
         rc = fail(SILO_GPU_ERR_HIP, "runs of the missing symbol: the two passes over the plane disagree");
      }
      if (rc == SILO_GPU_OK) {
         rc = silo_gpu_internal_sort_pairs(d_keys, d_ends, n_runs);  // by (sequence, start)
      }
   }
   (void)hipFree(d_count);
   if (status != hipSuccess || rc != SILO_GPU_OK) {
      (void)hipFree(d_keys);
      (void)hipFree(d_ends);
      HIP_TRY(status);
      return rc;
   }
   (void)hipFree(dev.extra);
   dev.extra = nullptr;
   store->device_bytes -= plane_bytes;
   store->device_bytes += slots * (sizeof(uint64_t) + sizeof(uint32_t));
   seqstore.d_missing_run_keys = d_keys;
   seqstore.d_missing_run_ends = d_ends;
   dev.missing_run_keys = d_keys;
   dev.missing_run_ends = d_ends;
   dev.n_missing_runs = static_cast<uint32_t>(n_runs);
   dev.kind[dev.missing_symbol] = PLANE_RUNS;
   return SILO_GPU_OK;
}

/// first[slice] = first run of the missing symbol whose sequence lies in slice `slice` of 2^ESCAPE_SLICE_SHIFT sequences or
/// beyond (the runs are sorted by sequence): one binary search per entry.
__global__ void k_run_slice_index(const uint64_t* __restrict__ run_keys, uint32_t n_runs, uint32_t slice_shift, uint32_t n_entries, uint32_t* __restrict__ first) {
   const uint32_t slice = blockIdx.x * blockDim.x + threadIdx.x;
   if (slice >= n_entries) {
      return;
   }
   uint32_t lo = 0, hi = n_runs;
   while (lo < hi) {
      const uint32_t mid = lo + (hi - lo) / 2;
      if ((static_cast<uint32_t>(run_keys[mid] >> 32) >> slice_shift) < slice) {
         lo = mid + 1;
      } else {
         hi = mid;
      }
   }
   first[slice] = lo;
}

/// A store with derived symbols counts, per scan, the rows of the filter inside a run of the missing symbol: where the runs
/// of every slice of sequences begin (k_scan_missing_runs keeps that slice of the filter in LDS).
int buildRunSliceIndex(silo_gpu_store* store, SeqStoreHost& seqstore) {
   SeqStoreHost::Layout& layout = seqstore.layout;
   if (!layout.has_implicit || layout.d_run_slice_first != nullptr) {
      return SILO_GPU_OK;
   }
   const SeqStoreDev& dev = seqstore.dev;
   if (dev.kind[dev.missing_symbol] != PLANE_RUNS) {
      return fail(SILO_GPU_ERR_INVALID_ARGUMENT, "a store with derived symbols keeps the missing symbol as runs");
   }
   layout.n_run_slices = (store->sequence_count + (1u << ESCAPE_SLICE_SHIFT) - 1) >> ESCAPE_SLICE_SHIFT;
   const uint32_t n_entries = layout.n_run_slices + 1;
   HIP_TRY(hipMalloc(&layout.d_run_slice_first, n_entries * sizeof(uint32_t)));
   k_run_slice_index<<<(n_entries + 255) / 256, 256>>>(dev.missing_run_keys, dev.n_missing_runs, ESCAPE_SLICE_SHIFT, n_entries, layout.d_run_slice_first);
   HIP_TRY(hipGetLastError());
   HIP_TRY(hipStreamSynchronize(nullptr));
   return SILO_GPU_OK;
}

int finalizeSeqStore(silo_gpu_store* store, SeqStoreHost& seqstore) {
   if (seqstore.layout.built) {
      return SILO_GPU_OK;
   }
   if (seqstore.dev.build_mode == BUILD_COUNT) {
      return fail(SILO_GPU_ERR_INVALID_ARGUMENT, "the sequence store is in the counting pass of a two-pass build: the encoding pass has to follow before finalize");
   }
   if (const int rc = ensureBuildPlanes(store, seqstore); rc != SILO_GPU_OK) {  // a store that never received a sequence: all-zero planes
      return rc;
   }
   uint32_t count = 0;
   HIP_TRY(hipMemcpy(&count, seqstore.d_sparse_count, sizeof(uint32_t), hipMemcpyDeviceToHost));
   count = std::min(count, seqstore.sparse_capacity);
   seqstore.sparse_sorted.resize(count);
   if (count > 0) {
      HIP_TRY(hipMemcpy(seqstore.sparse_sorted.data(), seqstore.d_sparse, static_cast<size_t>(count) * sizeof(uint64_t), hipMemcpyDeviceToHost));
      std::sort(seqstore.sparse_sorted.begin(), seqstore.sparse_sorted.end());
      // a replayed batch may have appended duplicates
      seqstore.sparse_sorted.erase(std::unique(seqstore.sparse_sorted.begin(), seqstore.sparse_sorted.end()), seqstore.sparse_sorted.end());
      count = static_cast<uint32_t>(seqstore.sparse_sorted.size());
      HIP_TRY(hipMemcpy(seqstore.d_sparse, seqstore.sparse_sorted.data(), static_cast<size_t>(count) * sizeof(uint64_t), hipMemcpyHostToDevice));
      HIP_TRY(hipMemcpy(seqstore.d_sparse_count, &count, sizeof(uint32_t), hipMemcpyHostToDevice));
   }
   seqstore.finalized = true;
   // the missing symbol first: its plane goes before the adaptive planes come (a lower peak), and only a store that keeps it as
   // runs may derive the most numerous symbol of a position (planLayout)
   if (const int rc = compactMissingPlane(store, seqstore); rc != SILO_GPU_OK) {
      return rc;
   }
   if (const int rc = buildLayout(store, seqstore); rc != SILO_GPU_OK) {
      return rc;
   }
   return buildRunSliceIndex(store, seqstore);
}
}  // namespace

int silo_gpu_store_build_pass(silo_gpu_store* store, uint32_t seqstore_id, int pass) {
   if (store == nullptr || seqstore_id >= store->seqstores.size() || (pass != 1 && pass != 2)) {
      return fail(SILO_GPU_ERR_INVALID_ARGUMENT, "silo_gpu_store_build_pass: bad arguments (pass 1 = counting, 2 = encoding)");
   }
   std::lock_guard<std::mutex> lock(store->mutex);
   HIP_TRY(hipSetDevice(store->device));
   SeqStoreHost& seqstore = store->seqstores[seqstore_id];
   SeqStoreDev& dev = seqstore.dev;
   if (seqstore.layout.built) {
      return fail(SILO_GPU_ERR_INVALID_ARGUMENT, "silo_gpu_store_build_pass: the sequence store is finalized");
   }
   const size_t n_counters = static_cast<size_t>(dev.positions) * dev.n_scan;
   if (pass == 1) {
      if (dev.scan != nullptr || dev.extra != nullptr || dev.build_mode != BUILD_PLANES) {
         return fail(SILO_GPU_ERR_INVALID_ARGUMENT, "silo_gpu_store_build_pass: the counting pass has to come before any sequence of the store");
      }
      if (seqstore.d_totals == nullptr) {
         HIP_TRY(hipMalloc(&seqstore.d_totals, std::max<size_t>(n_counters, 1) * sizeof(uint32_t)));
      }
      HIP_TRY(hipMemset(seqstore.d_totals, 0, std::max<size_t>(n_counters, 1) * sizeof(uint32_t)));
      HIP_TRY(hipStreamSynchronize(nullptr));
      seqstore.totals_ready = false;
      dev.enc_counts = seqstore.d_totals;
      dev.build_mode = BUILD_COUNT;
      // the missing symbol, where it is the store's only extra plane, is counted (and then written) as runs right away
      dev.runs_at_build = dev.n_extra == 1 && dev.kind[dev.missing_symbol] == PLANE_EXTRA && dev.index[dev.missing_symbol] == 0 && missingRunsOption(store) >= 0 ? 1 : 0;
      if (dev.runs_at_build != 0) {
         if (seqstore.d_run_count == nullptr) {
            HIP_TRY(hipMalloc(&seqstore.d_run_count, sizeof(unsigned long long)));
         }
         HIP_TRY(hipMemset(seqstore.d_run_count, 0, sizeof(unsigned long long)));
         HIP_TRY(hipStreamSynchronize(nullptr));
         dev.enc_run_count = seqstore.d_run_count;
      }
      return SILO_GPU_OK;
   }
   if (dev.build_mode != BUILD_COUNT) {
      return fail(SILO_GPU_ERR_INVALID_ARGUMENT, "silo_gpu_store_build_pass: the encoding pass follows the counting pass");
   }
   HIP_TRY(hipDeviceSynchronize());  // every count of the first pass has landed
   {  // the sparsely stored symbols the first pass counted (it stored none): room for all of them, the counter starts over
      uint32_t counted = 0;
      HIP_TRY(hipMemcpy(&counted, seqstore.d_sparse_count, sizeof(uint32_t), hipMemcpyDeviceToHost));
      if (counted > 0xFFFF0000u) {
         return fail(SILO_GPU_ERR_INVALID_ARGUMENT, "two-pass build: too many sparsely stored symbols");
      }
      if (const int rc = growSparse(seqstore, counted + 1u); rc != SILO_GPU_OK) {
         return rc;
      }
      const uint32_t zero = 0;
      HIP_TRY(hipMemcpy(seqstore.d_sparse_count, &zero, sizeof(uint32_t), hipMemcpyHostToDevice));
   }
   dev.build_mode = BUILD_PLANES;
   dev.enc_counts = nullptr;
   seqstore.totals_ready = true;
   const bool runs_counted = dev.runs_at_build != 0;
   dev.runs_at_build = 0;
   if (!reencodes(store, dev)) {
      seqstore.totals_ready = false;
      return SILO_GPU_OK;  // a store that keeps its identity planes: the second pass builds them the ordinary way
   }
   auto work = std::make_shared<SeqStoreHost::LayoutWork>();
   bool fits = false;
   unsigned long long n_runs = 0;
   if (runs_counted) {
      HIP_TRY(hipMemcpy(&n_runs, seqstore.d_run_count, sizeof(n_runs), hipMemcpyDeviceToHost));
   }
   // (the most numerous symbol of a position is derived only where the missing symbol is kept as runs)
   if (const int rc = planLayout(store, seqstore, *work, true, runs_counted && n_runs < (1ull << 32) && seqstore.rows_filled == store->sequence_count, &fits); rc != SILO_GPU_OK) {
      return rc;
   }
   if (!fits) {
      seqstore.totals_ready = false;
      return SILO_GPU_OK;
   }
   if (runs_counted) {  // the runs of the missing symbol: exactly as many slots as the first pass counted
      if (n_runs < (1ull << 32)) {
         const size_t slots = std::max<size_t>(n_runs, 1);
         hipError_t status = hipMalloc(&seqstore.d_missing_run_keys, slots * sizeof(uint64_t));
         status = status != hipSuccess ? status : hipMalloc(&seqstore.d_missing_run_ends, slots * sizeof(uint32_t));
         status = status != hipSuccess ? status : hipMemset(seqstore.d_run_count, 0, sizeof(unsigned long long));
         status = status != hipSuccess ? status : hipStreamSynchronize(nullptr);
         if (status != hipSuccess) {
            work->discard();
            HIP_TRY(status);
         }
         dev.enc_run_keys = seqstore.d_missing_run_keys;
         dev.enc_run_ends = seqstore.d_missing_run_ends;
         dev.enc_run_capacity = n_runs;
         dev.runs_at_build = 1;
      }
   }
   dev.enc_code_map = work->d_code_map;
   dev.enc_row_of = work->d_row_of;
   dev.enc_planes = work->d_planes;
   dev.enc_first = work->d_first;
   dev.enc_cursor = work->d_cursor;
   dev.enc_escapes = work->d_escapes;
   dev.build_mode = BUILD_ENCODE;
   seqstore.work = std::move(work);
   return SILO_GPU_OK;
}

int silo_gpu_store_build_mode(const silo_gpu_store* store, uint32_t seqstore_id) {
   if (store == nullptr || seqstore_id >= store->seqstores.size()) {
      return -1;
   }
   return static_cast<int>(store->seqstores[seqstore_id].dev.build_mode);
}

int silo_gpu_store_finalize(silo_gpu_store* store) {
   if (store == nullptr) {
      return fail(SILO_GPU_ERR_INVALID_ARGUMENT, "silo_gpu_store_finalize: null store");
   }
   std::lock_guard<std::mutex> lock(store->mutex);
   HIP_TRY(hipSetDevice(store->device));
   for (SeqStoreHost& seqstore : store->seqstores) {
      if (const int rc = finalizeSeqStore(store, seqstore); rc != SILO_GPU_OK) {
         return rc;
      }
   }
   return SILO_GPU_OK;
}

int silo_gpu_store_finalize_seqstore(silo_gpu_store* store, uint32_t seqstore_id) {
   if (store == nullptr || seqstore_id >= store->seqstores.size()) {
      return fail(SILO_GPU_ERR_INVALID_ARGUMENT, "silo_gpu_store_finalize_seqstore: bad arguments");
   }
   std::lock_guard<std::mutex> lock(store->mutex);
   HIP_TRY(hipSetDevice(store->device));
   return finalizeSeqStore(store, store->seqstores[seqstore_id]);
}

int silo_gpu_malloc(size_t bytes, void** out_dev) {
   if (out_dev == nullptr) {
      return fail(SILO_GPU_ERR_INVALID_ARGUMENT, "silo_gpu_malloc: null out pointer");
   }
   HIP_TRY(hipMalloc(out_dev, bytes));
   return SILO_GPU_OK;
}

void silo_gpu_free(void* dev_ptr) {
   (void)hipFree(dev_ptr);
}

}  // extern "C"

namespace {

/// A position range of one sequence store with the count tables of every filter of the launch.
struct ScanRange {
   const SeqStoreHost* seqstore;
   uint32_t pos_begin;
   uint32_t pos_end;
   uint32_t* counts[SILO_GPU_MAX_SCAN_BATCH];
};

/// The part of a range that lies in ONE run of its store's layout: what a launch takes.
struct ScanPiece {
   const uint64_t* planes;    // first plane row of the piece
   const uint8_t* code_map;   // of the piece's first position (mapped layouts); the row targets of its first row (one-hot rows); else nullptr
   uint32_t n_positions;      // one-hot rows: plane rows
   uint32_t target_base;      // one-hot rows: first position of the piece * n_scan
   uint32_t* counts[SILO_GPU_MAX_SCAN_BATCH];  // tables at the piece's first position
};

/// The four plane layouts the scan kernels are instantiated for.
enum ScanLayout { SCAN_2_PLANES = 0, SCAN_3_PLANES_MAPPED, SCAN_FULL_NUCLEOTIDE, SCAN_FULL_AMINO_ACID, SCAN_ONE_HOT_ROWS, N_SCAN_LAYOUTS };

ScanLayout layoutOfRun(const SeqStoreDev& dev, uint32_t bits, bool identity, bool one_hot) {
   if (one_hot) {
      return SCAN_ONE_HOT_ROWS;
   }
   if (!identity) {
      return bits == 2 ? SCAN_2_PLANES : SCAN_3_PLANES_MAPPED;
   }
   return dev.n_bits == 3 ? SCAN_FULL_NUCLEOTIDE : SCAN_FULL_AMINO_ACID;
}

/// Cuts the ranges along the runs of their stores; pieces[layout] collects what one kind of launch takes.
void cutIntoPieces(const std::vector<ScanRange>& ranges, uint32_t q_count, std::vector<ScanPiece> (&pieces)[N_SCAN_LAYOUTS]) {
   for (const ScanRange& range : ranges) {
      const SeqStoreHost& seqstore = *range.seqstore;
      const SeqStoreDev& dev = seqstore.dev;
      const auto add = [&](uint32_t begin, uint32_t end, uint32_t bits, bool identity, bool one_hot) {
         begin = std::max(begin, range.pos_begin);
         end = std::min(end, range.pos_end);
         if (begin >= end) {
            return;
         }
         ScanPiece piece{};
         const bool encoded = seqstore.layout.built && seqstore.layout.d_row_of != nullptr;
         const size_t first_row = encoded ? seqstore.layout.row_of[begin] : static_cast<size_t>(begin) * dev.n_bits;
         piece.planes = dev.planes + first_row * dev.row_words;
         piece.n_positions = end - begin;
         if (one_hot) {
            piece.code_map = reinterpret_cast<const uint8_t*>(seqstore.layout.d_row_target + first_row);
            piece.n_positions = seqstore.layout.row_of[end] - seqstore.layout.row_of[begin];
            piece.target_base = begin * dev.n_scan;
            if (piece.n_positions == 0) {
               return;  // positions whose only stored symbol is derived: no rows
            }
         } else if (!identity) {
            piece.code_map = seqstore.layout.d_code_map + static_cast<size_t>(begin) * CODE_MAP_STRIDE;
         }
         for (uint32_t q = 0; q < q_count; ++q) {
            piece.counts[q] = range.counts[q] + static_cast<size_t>(begin - range.pos_begin) * dev.n_scan;
         }
         pieces[layoutOfRun(dev, bits, identity, one_hot)].push_back(piece);
      };
      if (seqstore.layout.runs.empty()) {  // still the build-time planes (the totals scan inside finalize)
         add(0, dev.positions, dev.n_bits, true, false);
      }
      for (const SeqStoreHost::Run& run : seqstore.layout.runs) {
         add(run.begin, run.end, run.bits, run.identity, run.one_hot);
      }
   }
}

/// Event pairs around the plane-scan launches of this thread's last scan (SILO_GPU_TUNE_SCAN_TIMING); the events are
/// created once and reused.
struct ScanLaunchTiming {
   hipEvent_t start = nullptr;
   hipEvent_t stop = nullptr;
   silo_gpu_scan_timing entry{};
};
struct ScanTimingLog {
   std::vector<ScanLaunchTiming> launches;
   size_t used = 0;
};
ScanTimingLog& scanTimingLog() {
   thread_local ScanTimingLog log;
   return log;
}

/// With SILO_GPU_TUNE_SCAN_TIMING set: an entry of the thread's timing log with its start event recorded on `stream` (the
/// stream the launch that follows goes to); nullptr otherwise.  `bytes` = what the launch has to read, each byte once.
ScanLaunchTiming* startLaunchTiming(const char* kernel, uint64_t plane_rows, uint64_t bytes, uint32_t filters, uint32_t blocks, hipStream_t stream) {
   if (g_tune_scan_timing.load() != 1) {
      return nullptr;
   }
   ScanTimingLog& log = scanTimingLog();
   if (log.used == log.launches.size()) {
      ScanLaunchTiming fresh;
      if (hipEventCreate(&fresh.start) != hipSuccess || hipEventCreate(&fresh.stop) != hipSuccess) {
         (void)hipGetLastError();
         return nullptr;
      }
      log.launches.push_back(fresh);
   }
   ScanLaunchTiming* timing = &log.launches[log.used++];
   std::snprintf(timing->entry.kernel, sizeof(timing->entry.kernel), "%s", kernel);
   timing->entry.plane_rows = plane_rows;
   timing->entry.bytes = bytes;
   timing->entry.filters = filters;
   timing->entry.blocks = blocks;
   if (hipEventRecord(timing->start, stream) != hipSuccess) {
      (void)hipGetLastError();
      --log.used;
      return nullptr;
   }
   return timing;
}

void finishLaunchTiming(ScanLaunchTiming* timing, hipStream_t stream) {
   if (timing != nullptr) {
      (void)hipEventRecord(timing->stop, stream);
   }
}

/// Launches k_scan_sliced for the `q_count` filters and the pieces already entered in `batch` (planes, n_positions, counts).
template <int BITS, int NSYM, int KIND>
int launchSlicedScan(ScanBatchArgs& batch, uint32_t row_words, uint32_t q_count, hipStream_t hip_stream) {
   // words per thread: 8 for one filter over a layout of at most 5 counted symbols (2 or 3 planes x 4 chunks per position and
   // buffer), 4 otherwise (7 or 22 symbols; batches: Q filter tiles in registers).  SILO_GPU_TUNE_SCAN_VARIANT 10 / 12 force 4 / 8.
   const int variant = g_tune_scan_variant.load();
   constexpr bool CAN_BE_WIDE = BITS <= 3;
   bool wide = CAN_BE_WIDE && q_count == 1 && row_words >= SCAN_THREADS * 8;
   if (variant == 10) {
      wide = false;
   } else if (variant == 12 && CAN_BE_WIDE && q_count == 1) {
      wide = true;
   }
   const uint32_t tile_words = SCAN_THREADS * (wide ? 8 : 4);
   int positions_per_block = g_tune_rows_per_block.load();
   const uint32_t n_tiles = (row_words + tile_words - 1) / tile_words;
   // what the pipeline steps through: positions of BITS planes, or pairs of one-hot rows
   const auto units = [&](uint32_t r) { return KIND == KIND_ROWS ? (batch.n_positions[r] + 1u) / 2u : batch.n_positions[r]; };
   uint64_t total_positions = 0;
   for (uint32_t r = 0; r < batch.n_ranges; ++r) {
      total_positions += units(r);
   }
   if (positions_per_block <= 0) {
      // 2 or 3 planes per position: 128 positions per block while that still leaves >= 4096 blocks, else 64; the 5 identity
      // planes of amino acids: 12 (60 plane rows) — profiles/r01_scan_variants.md
      // (a block re-reads its filter tile — one plane row's worth — whatever it scans, so fewer positions per block cost
      // 1 / (positions x planes) more bytes; too few blocks leave the chip idle at the launch's tail)
      positions_per_block = 12;
      if constexpr (BITS <= 3) {
         positions_per_block = 128;
         while (positions_per_block > 32 && static_cast<uint64_t>(n_tiles) * ((total_positions + positions_per_block - 1) / positions_per_block) < 12288) {
            positions_per_block /= 2;
         }
      }
   }
   positions_per_block += positions_per_block & 1;  // the pipeline works on pairs of positions
   batch.first_unit[0] = 0;
   for (uint32_t r = 0; r < batch.n_ranges; ++r) {
      batch.first_unit[r + 1] = batch.first_unit[r] + n_tiles * ((units(r) + positions_per_block - 1) / positions_per_block);
   }
   const dim3 grid(batch.first_unit[batch.n_ranges]);
   ScanLaunchTiming* timing = nullptr;
   if (g_tune_scan_timing.load() == 1) {
      uint64_t plane_rows = 0;
      for (uint32_t r = 0; r < batch.n_ranges; ++r) {
         plane_rows += KIND == KIND_ROWS ? batch.n_positions[r] : static_cast<uint64_t>(batch.n_positions[r]) * BITS;
      }
      char name[64];
      std::snprintf(name, sizeof(name), "k_scan_sliced<%d, %d, %d, %u, %d>", BITS, NSYM, wide ? 8 : 4, wide ? 1u : std::min(q_count, 8u), KIND);
      timing = startLaunchTiming(name, plane_rows, (plane_rows + q_count) * row_words * sizeof(uint64_t), q_count, grid.x, hip_stream);
   }
#define SILO_LAUNCH_SLICED(WPT, Q) \
   k_scan_sliced<BITS, NSYM, WPT, Q, KIND><<<grid, SCAN_THREADS, 0, hip_stream>>>(batch, row_words, positions_per_block, n_tiles)
   if (wide) {
      if constexpr (CAN_BE_WIDE) {
         SILO_LAUNCH_SLICED(8, 1);
      }
   } else {
      switch (q_count) {
         case 1: SILO_LAUNCH_SLICED(4, 1); break;
         case 2: SILO_LAUNCH_SLICED(4, 2); break;
         case 3: SILO_LAUNCH_SLICED(4, 3); break;
         case 4: SILO_LAUNCH_SLICED(4, 4); break;
         default:
            if constexpr (NSYM <= 5) {  // 5..8 filters: layouts of at most 5 counted symbols (the others go in groups of 4)
               switch (q_count) {
                  case 5: SILO_LAUNCH_SLICED(4, 5); break;
                  case 6: SILO_LAUNCH_SLICED(4, 6); break;
                  case 7: SILO_LAUNCH_SLICED(4, 7); break;
                  default: SILO_LAUNCH_SLICED(4, 8); break;
               }
            } else {
               return fail(SILO_GPU_ERR_INVALID_ARGUMENT, "launchSlicedScan: more than 4 filters in one pass over a 7- or 22-symbol layout");
            }
      }
   }
#undef SILO_LAUNCH_SLICED
   HIP_TRY(hipGetLastError());
   finishLaunchTiming(timing, hip_stream);
   return SILO_GPU_OK;
}

/// Launches k_scan_gather (one wave per POSG positions) for the pieces in `batch`; grid.y = filter.
template <int BITS, int NSYM, int POSG, int KIND>
int launchGatherScan(ScanBatchArgs& batch, const uint32_t* sector_index, uint32_t stride, uint32_t row_words, uint32_t q_count, hipStream_t hip_stream) {
   batch.first_unit[0] = 0;
   for (uint32_t r = 0; r < batch.n_ranges; ++r) {
      batch.first_unit[r + 1] = batch.first_unit[r] + (batch.n_positions[r] + POSG - 1) / POSG;
   }
   const uint32_t waves = batch.first_unit[batch.n_ranges];
   k_scan_gather<BITS, NSYM, POSG, KIND><<<dim3((waves + 3) / 4, q_count), 256, 0, hip_stream>>>(batch, sector_index, stride, row_words);
   HIP_TRY(hipGetLastError());
   return SILO_GPU_OK;
}

/// Device scratch of a scan: per filter the counters of the prepare step (TWO sets: a scan uses one and zeroes the other for
/// the next scan on this block, so no fill launch is needed), the list of sector indexes of the sparse-filter routing, and
/// the private tables of a scan with derived symbols.  Blocks are pooled; a block is handed out again only once the event
/// recorded after its last use has completed, whatever stream that use was on.
struct SparseScratch {
   int device = 0;
   uint32_t capacity = 0;  // sectors per filter
   uint32_t* counters[2] = {nullptr, nullptr};  // [SILO_GPU_MAX_SCAN_BATCH * SPARSE_COUNTER_STRIDE] each
   uint32_t set = 0;                            // the counter set of the current use
   uint32_t* sector_index = nullptr;            // [SILO_GPU_MAX_SCAN_BATCH][capacity]
   uint32_t* tables = nullptr;                  // private count tables (scans with derived symbols)
   size_t table_words = 0;
   hipEvent_t last_use = nullptr;
   bool in_flight = false;  // handed out and not yet released
};

std::mutex g_sparse_scratch_mutex;
std::vector<SparseScratch*> g_sparse_scratch;

int acquireSparseScratch(int device, uint32_t capacity, size_t table_words, SparseScratch** out) {
   {
      std::lock_guard<std::mutex> lock(g_sparse_scratch_mutex);
      for (SparseScratch* block : g_sparse_scratch) {
         if (!block->in_flight && block->device == device && block->capacity >= capacity && block->table_words >= table_words &&
             hipEventQuery(block->last_use) == hipSuccess) {
            block->in_flight = true;
            block->set ^= 1u;
            *out = block;
            return SILO_GPU_OK;
         }
      }
   }
   auto block = std::make_unique<SparseScratch>();
   block->device = device;
   block->capacity = capacity;
   block->table_words = std::max<size_t>(table_words, size_t{1} << 20);
   void* memory = nullptr;
   // one allocation: the private tables, the index lists, then the two counter sets (zeroed here, by the scans from then on)
   const size_t counter_words = static_cast<size_t>(SILO_GPU_MAX_SCAN_BATCH) * SPARSE_COUNTER_STRIDE;
   const size_t bytes = (block->table_words + static_cast<size_t>(SILO_GPU_MAX_SCAN_BATCH) * capacity + 2 * counter_words) * sizeof(uint32_t);
   HIP_TRY(hipMalloc(&memory, bytes));
   block->tables = static_cast<uint32_t*>(memory);
   block->sector_index = block->tables + block->table_words;
   block->counters[0] = block->sector_index + static_cast<size_t>(SILO_GPU_MAX_SCAN_BATCH) * capacity;
   block->counters[1] = block->counters[0] + counter_words;
   hipError_t status = hipMemset(block->counters[0], 0, 2 * counter_words * sizeof(uint32_t));
   status = status != hipSuccess ? status : hipStreamSynchronize(nullptr);  // the fill is only enqueued; the scans run on other streams
   status = status != hipSuccess ? status : hipEventCreateWithFlags(&block->last_use, hipEventDisableTiming);
   if (status != hipSuccess) {
      (void)hipFree(memory);
      return fail(SILO_GPU_ERR_HIP, "scan scratch: " + std::string(hipGetErrorString(status)));
   }
   block->in_flight = true;
   std::lock_guard<std::mutex> lock(g_sparse_scratch_mutex);
   g_sparse_scratch.push_back(block.get());
   *out = block.release();
   return SILO_GPU_OK;
}

void releaseSparseScratch(SparseScratch* block, hipStream_t stream) {
   (void)hipEventRecord(block->last_use, stream);
   std::lock_guard<std::mutex> lock(g_sparse_scratch_mutex);
   block->in_flight = false;
}



/// Side streams (and the events that tie them to the caller's) per host thread.  The escape pass is a stream of keys, random
/// filter lookups and atomics — latency-bound — and adds to the same count tables as the plane scans, which are
/// bandwidth-bound, so it runs beside them.  Never destroyed (thread exit may come after the HIP runtime has shut down).
constexpr int N_SIDE_STREAMS = 2;  // the escape pass: [0] at the lowest stream priority, [1] at the default one (SILO_GPU_TUNE_SIDE_STREAM)
struct SideStreams {
   hipStream_t stream[N_SIDE_STREAMS] = {nullptr, nullptr};
   hipEvent_t fork[2] = {nullptr, nullptr};
   hipEvent_t join[N_SIDE_STREAMS] = {nullptr, nullptr};
   bool used[N_SIDE_STREAMS] = {false, false};
   bool tried = false;
   bool ok = false;
};

SideStreams* sideStreams() {
   thread_local SideStreams side;
   if (!side.tried) {
      side.tried = true;
      side.ok = true;
      int least = 0, greatest = 0;
      (void)hipDeviceGetStreamPriorityRange(&least, &greatest);
      for (int k = 0; k < N_SIDE_STREAMS; ++k) {
         side.ok = side.ok && hipStreamCreateWithPriority(&side.stream[k], hipStreamNonBlocking, k == 0 ? least : 0) == hipSuccess &&
                   hipEventCreateWithFlags(&side.join[k], hipEventDisableTiming) == hipSuccess;
      }
      for (int k = 0; k < 2; ++k) {
         side.ok = side.ok && hipEventCreateWithFlags(&side.fork[k], hipEventDisableTiming) == hipSuccess;
      }
      if (!side.ok) {
         (void)hipGetLastError();
      }
   }
   return side.ok ? &side : nullptr;
}

/// Makes side stream `k` wait for everything queued on `hip_stream` so far (through fork event `fork_index`).
int forkSide(SideStreams* side, int k, int fork_index, hipStream_t hip_stream, bool record) {
   if (record) {
      HIP_TRY(hipEventRecord(side->fork[fork_index], hip_stream));
   }
   HIP_TRY(hipStreamWaitEvent(side->stream[k], side->fork[fork_index], 0));
   side->used[k] = true;
   return SILO_GPU_OK;
}

/// Makes `hip_stream` wait for every side stream that was used since the last join.
int joinSides(hipStream_t hip_stream) {
   SideStreams* side = sideStreams();
   if (side == nullptr) {
      return SILO_GPU_OK;
   }
   for (int k = 0; k < N_SIDE_STREAMS; ++k) {
      if (side->used[k]) {
         side->used[k] = false;
         HIP_TRY(hipEventRecord(side->join[k], side->stream[k]));
         HIP_TRY(hipStreamWaitEvent(hip_stream, side->join[k], 0));
      }
   }
   return SILO_GPU_OK;
}

/// Fills the piece-dependent part of a launch descriptor from pieces[first, first + n).
void enterPieces(ScanBatchArgs& batch, const std::vector<ScanPiece>& pieces, size_t first, uint32_t n, uint32_t first_filter, uint32_t n_filters) {
   batch.n_ranges = n;
   for (uint32_t r = 0; r < n; ++r) {
      const ScanPiece& piece = pieces[first + r];
      batch.planes[r] = piece.planes;
      batch.code_map[r] = piece.code_map;
      batch.target_base[r] = piece.target_base;
      batch.n_positions[r] = piece.n_positions;
      for (uint32_t q = 0; q < n_filters; ++q) {
         batch.counts[r][q] = piece.counts[first_filter + q];
      }
   }
}

/// The dense kernels for `q_count` filters over the pieces of every layout: at most SCAN_MAX_RANGES pieces and 8 (layouts
/// of 3 or 5 counted symbols) or 4 (7 or 22) filters per launch.  sparse_sectors carries the routing counters (or nullptr).
int scanPiecesDense(
   const std::vector<ScanPiece> (&pieces)[N_SCAN_LAYOUTS], const SeqStoreDev& any_store, const uint64_t* const* filters, uint32_t q_count,
   const uint32_t* sparse_sectors, uint32_t sparse_capacity, hipStream_t hip_stream
) {
   // (running the plane scans of a query's smaller layouts on side streams beside the largest one was tried: no gain, the
   // launches are bandwidth-bound together — profiles/r02_amino_acid.md)
   for (int layout = 0; layout < N_SCAN_LAYOUTS; ++layout) {
      const std::vector<ScanPiece>& list = pieces[layout];
      const uint32_t filters_per_pass = layout == SCAN_2_PLANES || layout == SCAN_FULL_NUCLEOTIDE || layout == SCAN_ONE_HOT_ROWS ? SILO_GPU_MAX_SCAN_BATCH : 4;
      for (size_t first_piece = 0; first_piece < list.size(); first_piece += SCAN_MAX_RANGES) {
         const uint32_t n_pieces = static_cast<uint32_t>(std::min<size_t>(SCAN_MAX_RANGES, list.size() - first_piece));
         for (uint32_t first = 0; first < q_count; first += filters_per_pass) {
            const uint32_t n = std::min<uint32_t>(filters_per_pass, q_count - first);
            ScanBatchArgs batch{};
            batch.sparse_sectors = sparse_sectors != nullptr ? sparse_sectors + first * SPARSE_COUNTER_STRIDE : nullptr;
            batch.sparse_capacity = sparse_capacity;
            batch.out_symbols = any_store.n_scan;
            for (uint32_t q = 0; q < n; ++q) {
               batch.filters[q] = filters[first + q];
            }
            enterPieces(batch, list, first_piece, n_pieces, first, n);
            int rc = SILO_GPU_OK;
            switch (layout) {
               case SCAN_2_PLANES: rc = launchSlicedScan<2, 3, KIND_MAPPED>(batch, any_store.row_words, n, hip_stream); break;
               case SCAN_3_PLANES_MAPPED: rc = launchSlicedScan<3, 7, KIND_MAPPED>(batch, any_store.row_words, n, hip_stream); break;
               case SCAN_FULL_NUCLEOTIDE: rc = launchSlicedScan<3, 5, KIND_IDENTITY>(batch, any_store.row_words, n, hip_stream); break;
               case SCAN_ONE_HOT_ROWS: rc = launchSlicedScan<2, 2, KIND_ROWS>(batch, any_store.row_words, n, hip_stream); break;
               default: rc = launchSlicedScan<5, 22, KIND_IDENTITY>(batch, any_store.row_words, n, hip_stream); break;
            }
            if (rc != SILO_GPU_OK) {
               return rc;
            }
         }
      }
   }
   return SILO_GPU_OK;
}

/// The rows the code planes do not carry: one pass over the escape keys of every range, for all filters (dense and
/// sparse alike: the gather reads the same planes).
int scanEscapes(const std::vector<ScanRange>& ranges, const uint64_t* const* filters, uint32_t q_count, hipStream_t hip_stream) {
   // the ranges whose stores have slice-major keys go ESCAPE_MAX_RANGES at a time into one launch of k_scan_escapes_sliced
   EscapeSliceArgs sliced{};
   uint32_t n_sliced = 0;
   uint32_t most_keys = 0;  // of one (range, slice)
   uint64_t total_keys = 0, total_positions = 0;  // of the ranges of the launch
   const auto launchSliced = [&]() -> int {
      if (n_sliced == 0) {
         return SILO_GPU_OK;
      }
      static std::once_flag lds_once;
      std::call_once(lds_once, [] {  // filter slices + counter windows: beyond what a kernel may ask for by default
         (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_scan_escapes_sliced<1>), hipFuncAttributeMaxDynamicSharedMemorySize, escapeLdsBytes<1>());
         (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_scan_escapes_sliced<1, false>), hipFuncAttributeMaxDynamicSharedMemorySize, escapeLdsBytes<1>());
         (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_scan_escapes_sliced<2>), hipFuncAttributeMaxDynamicSharedMemorySize, escapeLdsBytes<2>());
         (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_scan_escapes_sliced<4>), hipFuncAttributeMaxDynamicSharedMemorySize, escapeLdsBytes<4>());
         (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_scan_escapes_sliced<8>), hipFuncAttributeMaxDynamicSharedMemorySize, escapeLdsBytes<8>());
      });
      const uint32_t per_block = q_count <= 1 ? 1 : (q_count <= 2 ? 2 : (q_count <= 4 ? 4 : 8));  // filters per pass over the keys
      const uint32_t chunk_keys = ESCAPE_SLICE_THREADS * (per_block >= 8 ? escapeKeysInFlight<8>() : (per_block >= 4 ? escapeKeysInFlight<4>() : escapeKeysInFlight<1>()));
      // keys per block: as many as mostly fall into the block's window of counters (3/4 of it at the average density of keys
      // along the positions; a key beyond it still counts, with an atomic of its own), at most ESCAPE_CHUNKS_PER_BLOCK chunks
      const uint32_t window = per_block >= 8 ? escapeWindow<8>() : (per_block >= 4 ? escapeWindow<4>() : (per_block >= 2 ? escapeWindow<2>() : escapeWindow<1>()));
      const double keys_per_counter = static_cast<double>(total_keys) / std::max<double>(1.0, static_cast<double>(total_positions) * sliced.n_slices * sliced.out_symbols);
      const double fitting = 0.75 * window * keys_per_counter;
      const uint32_t block_keys = static_cast<uint32_t>(std::min<double>(chunk_keys * ESCAPE_CHUNKS_PER_BLOCK, std::max<double>(2048.0, fitting))) & ~1u;
      sliced.block_keys = block_keys;
      const dim3 grid((most_keys + block_keys - 1) / block_keys, sliced.n_slices * n_sliced, (q_count + per_block - 1) / per_block);
      char name[64];
      std::snprintf(name, sizeof(name), "k_scan_escapes_sliced<%u, true>", per_block);
      // bytes: the keys once per pass of `per_block` filters, plus a 16 KiB filter slice per block and filter
      ScanLaunchTiming* timing = startLaunchTiming(
         name, 0, total_keys * sizeof(uint64_t) * grid.z + static_cast<uint64_t>(grid.x) * grid.y * q_count * ESCAPE_SLICE_WORDS32 * sizeof(uint32_t), q_count,
         grid.x * grid.y * grid.z, hip_stream
      );
      switch (per_block) {
         case 1:
            if (g_tune_scan_variant.load() == 30) {
               k_scan_escapes_sliced<1, false><<<grid, ESCAPE_SLICE_THREADS, escapeLdsBytes<1>(), hip_stream>>>(sliced, q_count);
            } else {
               k_scan_escapes_sliced<1><<<grid, ESCAPE_SLICE_THREADS, escapeLdsBytes<1>(), hip_stream>>>(sliced, q_count);
            }
            break;
         case 2: k_scan_escapes_sliced<2><<<grid, ESCAPE_SLICE_THREADS, escapeLdsBytes<2>(), hip_stream>>>(sliced, q_count); break;
         case 4: k_scan_escapes_sliced<4><<<grid, ESCAPE_SLICE_THREADS, escapeLdsBytes<4>(), hip_stream>>>(sliced, q_count); break;
         default: k_scan_escapes_sliced<8><<<grid, ESCAPE_SLICE_THREADS, escapeLdsBytes<8>(), hip_stream>>>(sliced, q_count); break;
      }
      HIP_TRY(hipGetLastError());
      finishLaunchTiming(timing, hip_stream);
      n_sliced = 0;
      most_keys = 0;
      total_keys = 0;
      total_positions = 0;
      return SILO_GPU_OK;
   };
   for (const ScanRange& range : ranges) {
      const SeqStoreHost::Layout& layout = range.seqstore->layout;
      if (!layout.built || layout.d_escapes == nullptr) {
         continue;
      }
      const uint32_t begin = layout.escape_first[range.pos_begin];
      const uint32_t count = layout.escape_first[range.pos_end] - begin;
      if (count == 0) {
         continue;
      }
      if (layout.d_escapes_sliced != nullptr && g_tune_side_stream.load() != 3) {  // the slice-major keys, a slice of the filter in LDS
         if (n_sliced == ESCAPE_MAX_RANGES || (n_sliced != 0 && sliced.n_slices != layout.n_slices)) {
            if (const int rc = launchSliced(); rc != SILO_GPU_OK) {
               return rc;
            }
         }
         sliced.row_words = range.seqstore->dev.row_words;
         sliced.n_slices = layout.n_slices;
         sliced.out_symbols = range.seqstore->dev.n_scan;
         EscapeSliceArgs::Range& entry = sliced.ranges[n_sliced++];
         entry.keys = layout.d_escapes_sliced;
         entry.slice_first = layout.d_slice_first;
         entry.positions = range.seqstore->dev.positions;
         entry.pos_begin = range.pos_begin;
         entry.pos_end = range.pos_end;
         for (uint32_t q = 0; q < q_count; ++q) {
            sliced.filters[q] = filters[q];
            entry.counts[q] = range.counts[q];
         }
         total_keys += count;
         total_positions += range.pos_end - range.pos_begin;
         const size_t stride = static_cast<size_t>(entry.positions) + 1;
         for (uint32_t slice = 0; slice < layout.n_slices; ++slice) {
            const uint32_t keys = layout.slice_first[slice * stride + range.pos_end] - layout.slice_first[slice * stride + range.pos_begin];
            most_keys = std::max(most_keys, keys + 1u);  // (chunks start at an even key index)
         }
         continue;
      }
      ScanBatchArgs escapes{};
      escapes.out_symbols = range.seqstore->dev.n_scan;
      for (uint32_t q = 0; q < q_count; ++q) {
         escapes.filters[q] = filters[q];
         escapes.counts[0][q] = range.counts[q];
      }
      const uint32_t keys_per_block = 256 * ESCAPE_KEYS_PER_THREAD;
      k_scan_escapes<<<dim3((count + keys_per_block - 1) / keys_per_block, q_count), 256, 0, hip_stream>>>(
         layout.d_escapes + begin, count, escapes, range.pos_begin
      );
      HIP_TRY(hipGetLastError());
   }
   return launchSliced();
}

/// The private tables of a scan with derived symbols and what its extra passes read, DERIVED_MAX_RANGES ranges at a time.
struct DerivedPlan {
   std::vector<DerivedArgs> launches;     // ranges [16 k, 16 k + 16) of the scan
   std::vector<ScanRange> private_ranges;  // the ranges with their count tables replaced by the private ones
   std::vector<std::array<uint64_t, DERIVED_MAX_RANGES>> run_counts;  // [launch][range] runs of the missing symbol of the range's store (for the timing log)
   size_t table_words = 0;
   uint32_t most_positions = 0;  // of a range with derived symbols
};

/// Lays the private tables of `ranges` out (offsets only: `tables` may still be null) .
void planDerived(const silo_gpu_store* store, const std::vector<ScanRange>& ranges, const uint64_t* const* filters, uint32_t q_count, DerivedPlan& plan) {
   plan.private_ranges = ranges;
   plan.launches.assign((ranges.size() + DERIVED_MAX_RANGES - 1) / DERIVED_MAX_RANGES, DerivedArgs{});
   plan.run_counts.assign(plan.launches.size(), {});
   size_t offset = 0;
   for (size_t r = 0; r < ranges.size(); ++r) {
      const ScanRange& range = ranges[r];
      const SeqStoreHost& seqstore = *range.seqstore;
      DerivedArgs& launch = plan.launches[r / DERIVED_MAX_RANGES];
      DerivedRange& entry = launch.ranges[launch.n_ranges++];
      const uint32_t n = range.pos_end - range.pos_begin;
      entry.n_positions = n;
      entry.n_scan = seqstore.dev.n_scan;
      entry.pos_begin = range.pos_begin;
      entry.stride = static_cast<uint32_t>((static_cast<size_t>(n) * seqstore.dev.n_scan + n + 1 + n + 3) / 4 * 4);
      entry.scratch = reinterpret_cast<uint32_t*>(offset * sizeof(uint32_t));  // + the scratch block's tables (bindDerived)
      offset += static_cast<size_t>(entry.stride) * q_count;
      if (seqstore.layout.has_implicit) {
         plan.run_counts[r / DERIVED_MAX_RANGES][launch.n_ranges - 1] = seqstore.dev.n_missing_runs;
         entry.code_map = seqstore.layout.d_code_map;
         entry.run_keys = seqstore.dev.missing_run_keys;
         entry.run_ends = seqstore.dev.missing_run_ends;
         entry.run_slice_first = seqstore.layout.d_run_slice_first;
         launch.n_run_slices = seqstore.layout.n_run_slices;
         entry.sparse_keys = seqstore.d_sparse;
         const auto lo = std::lower_bound(seqstore.sparse_sorted.begin(), seqstore.sparse_sorted.end(), static_cast<uint64_t>(range.pos_begin) << 37);
         const auto hi = std::lower_bound(lo, seqstore.sparse_sorted.end(), static_cast<uint64_t>(range.pos_end) << 37);
         entry.sparse_begin = static_cast<uint32_t>(lo - seqstore.sparse_sorted.begin());
         entry.sparse_end = static_cast<uint32_t>(hi - seqstore.sparse_sorted.begin());
         plan.most_positions = std::max(plan.most_positions, n);
      }
      for (uint32_t q = 0; q < q_count; ++q) {
         entry.caller_counts[q] = range.counts[q];
         launch.filters[q] = filters[q];
      }
      launch.row_words = store->row_words;
   }
   plan.table_words = offset;
}

/// The tables get their place in the scratch block; the private ranges point at them.
void bindDerived(DerivedPlan& plan, const SparseScratch& scratch, uint32_t q_count) {
   size_t r = 0;
   for (DerivedArgs& launch : plan.launches) {
      launch.counters = scratch.counters[scratch.set];
      for (uint32_t k = 0; k < launch.n_ranges; ++k, ++r) {
         DerivedRange& entry = launch.ranges[k];
         entry.scratch = scratch.tables + reinterpret_cast<size_t>(entry.scratch) / sizeof(uint32_t);
         for (uint32_t q = 0; q < q_count; ++q) {
            plan.private_ranges[r].counts[q] = entry.scratch + static_cast<size_t>(q) * entry.stride;
         }
      }
   }
}

/// Rows of the filters without a valid symbol, per position: the runs of the missing symbol and the sparse keys (ambiguity codes).
int scanRowsWithoutSymbol(DerivedPlan& plan, uint32_t q_count, hipStream_t hip_stream) {
   for (DerivedArgs& launch : plan.launches) {
      bool any = false;
      for (uint32_t k = 0; k < launch.n_ranges; ++k) {
         any = any || launch.ranges[k].code_map != nullptr;
      }
      if (!any) {
         continue;
      }
      // the diff of a range in LDS beside the filter slice, while it fits
      const size_t lds_bytes = (ESCAPE_SLICE_WORDS32 + static_cast<size_t>(plan.most_positions) + 1) * sizeof(uint32_t);
      const bool lds_diff = lds_bytes <= 152 * 1024;
      static std::once_flag lds_once;
      std::call_once(lds_once, [] {
         (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_scan_missing_runs<true>), hipFuncAttributeMaxDynamicSharedMemorySize, 152 * 1024);
      });
      // one block per CU fits (its LDS): about one round of the 256 CUs over all (slice, range, filter)
      const uint32_t run_units = std::max<uint32_t>(1, launch.n_run_slices * launch.n_ranges * q_count);
      const dim3 run_grid(std::min<uint32_t>(8, std::max<uint32_t>(1, 240 / run_units)), launch.n_run_slices * launch.n_ranges, q_count);
      uint64_t run_bytes = 0, sparse_bytes = 0;
      for (uint32_t k = 0; k < launch.n_ranges; ++k) {
         if (launch.ranges[k].code_map != nullptr) {
            run_bytes += plan.run_counts[&launch - plan.launches.data()][k] * (sizeof(uint64_t) + sizeof(uint32_t));
            sparse_bytes += static_cast<uint64_t>(launch.ranges[k].sparse_end - launch.ranges[k].sparse_begin) * sizeof(uint64_t);
         }
      }
      ScanLaunchTiming* run_timing = startLaunchTiming(lds_diff ? "k_scan_missing_runs<true>" : "k_scan_missing_runs<false>", 0, run_bytes * q_count, q_count, run_grid.x * run_grid.y * run_grid.z, hip_stream);
      if (lds_diff) {
         k_scan_missing_runs<true><<<run_grid, DERIVED_THREADS, lds_bytes, hip_stream>>>(launch);
      } else {
         k_scan_missing_runs<false><<<run_grid, DERIVED_THREADS, ESCAPE_SLICE_WORDS32 * sizeof(uint32_t), hip_stream>>>(launch);
      }
      HIP_TRY(hipGetLastError());
      finishLaunchTiming(run_timing, hip_stream);
      launch.first_unit[0] = 0;
      for (uint32_t k = 0; k < launch.n_ranges; ++k) {
         const uint32_t keys = launch.ranges[k].code_map != nullptr ? launch.ranges[k].sparse_end - launch.ranges[k].sparse_begin : 0;
         launch.first_unit[k + 1] = launch.first_unit[k] + (keys + 256 * SPARSE_KEYS_PER_THREAD - 1) / (256 * SPARSE_KEYS_PER_THREAD);
      }
      if (launch.first_unit[launch.n_ranges] != 0) {
         ScanLaunchTiming* sparse_timing = startLaunchTiming("k_count_sparse_keys", 0, sparse_bytes * q_count, q_count, launch.first_unit[launch.n_ranges] * q_count, hip_stream);
         k_count_sparse_keys<<<dim3(launch.first_unit[launch.n_ranges], q_count), 256, 0, hip_stream>>>(launch);
         HIP_TRY(hipGetLastError());
         finishLaunchTiming(sparse_timing, hip_stream);
      }
   }
   return SILO_GPU_OK;
}

/// The derived counts, and the private tables into the caller's.
int finishDerived(DerivedPlan& plan, uint32_t q_count, hipStream_t hip_stream) {
   for (DerivedArgs& launch : plan.launches) {
      launch.first_unit[0] = 0;
      for (uint32_t k = 0; k < launch.n_ranges; ++k) {
         launch.first_unit[k + 1] = launch.first_unit[k] + (launch.ranges[k].n_positions + DERIVED_THREADS - 1) / DERIVED_THREADS;
      }
      if (launch.first_unit[launch.n_ranges] != 0) {
         k_finish_scan<<<dim3(launch.first_unit[launch.n_ranges], q_count), DERIVED_THREADS, 0, hip_stream>>>(launch);
         HIP_TRY(hipGetLastError());
      }
   }
   return SILO_GPU_OK;
}

/// The passes beside the plane scans.  A store with a row for every stored symbol: the escape keys on side stream 0 (lowest
/// priority), beside plane scans that take milliseconds.  A scan with derived symbols has few plane rows left and its escape
/// pass is as long as its plane scans — both stream at the memory's rate and gain nothing from sharing it —, so the escape
/// pass stays on the caller's stream in front of the plane scans, and the side stream (default priority) takes the passes
/// that are bound by latency and LDS, not by bandwidth: the runs of the missing symbol and the sparse keys.  Forked behind
/// everything already queued on `hip_stream` (the filters are complete, the tables zeroed), joined by joinSides before
/// anything reads the tables.  SILO_GPU_TUNE_SIDE_STREAM: 0 as described, 1 side stream of default priority, 2 everything on the caller's stream.
int forkSidePasses(const std::vector<ScanRange>& ranges, const uint64_t* const* filters, uint32_t q_count, DerivedPlan* derived, hipStream_t hip_stream) {
   bool any_escapes = false;
   for (const ScanRange& range : ranges) {
      const SeqStoreHost::Layout& layout = range.seqstore->layout;
      any_escapes = any_escapes || (layout.built && layout.d_escapes != nullptr && layout.escape_first[range.pos_end] != layout.escape_first[range.pos_begin]);
   }
   if (!any_escapes && derived == nullptr) {
      return SILO_GPU_OK;
   }
   const int mode = g_tune_side_stream.load();
   SideStreams* side = mode == 2 ? nullptr : sideStreams();
   hipStream_t stream = hip_stream;
   if (side != nullptr) {
      const int k = mode == 1 || derived != nullptr ? 1 : 0;
      if (const int rc = forkSide(side, k, 0, hip_stream, true); rc != SILO_GPU_OK) {
         return rc;
      }
      stream = side->stream[k];
   }
   if (derived != nullptr) {
      if (const int rc = scanRowsWithoutSymbol(*derived, q_count, stream); rc != SILO_GPU_OK) {
         return rc;
      }
      return any_escapes ? scanEscapes(ranges, filters, q_count, mode == 1 ? stream : hip_stream) : SILO_GPU_OK;
   }
   return scanEscapes(ranges, filters, q_count, stream);
}

/// Scan of up to SILO_GPU_MAX_SCAN_BATCH filters over position ranges of sequence stores of one alphabet, with the
/// sparse-filter routing (K1s) around the dense kernels: every filter is compacted ONCE for all ranges, the dense
/// kernels skip the sparse ones, the gather kernel serves them.  All decisions are taken on the device.  Where a store
/// derives the most numerous symbol of its positions the kernels count into private tables and k_finish_scan completes them.
int scanRanges(
   const silo_gpu_store* store, const std::vector<ScanRange>& caller_ranges, const uint64_t* const* filters, uint32_t q_count, hipStream_t hip_stream
) {
   const SeqStoreDev& any_store = caller_ranges.front().seqstore->dev;
   const bool nucleotide = any_store.n_bits == 3 && any_store.n_scan == 5;
   if (!nucleotide && !(any_store.n_bits == 5 && any_store.n_scan == 22)) {
      return fail(SILO_GPU_ERR_INVALID_ARGUMENT, "mutations scan: unsupported set of scan symbols (5 nucleotide or 22 amino-acid symbols)");
   }
   if (any_store.row_words < SCAN_THREADS * 4) {
      // short rows: one wave per position over the identity planes (such stores keep them), one filter and one range at a time
      for (const ScanRange& range : caller_ranges) {
         const SeqStoreDev& dev = range.seqstore->dev;
         const uint32_t n_positions = range.pos_end - range.pos_begin;
         const uint32_t waves = std::min<uint32_t>(n_positions, 256u * 32u);
         const uint32_t blocks = (waves + 3) / 4;
         const uint64_t* planes = dev.planes + static_cast<size_t>(range.pos_begin) * dev.n_bits * dev.row_words;
         for (uint32_t q = 0; q < q_count; ++q) {
            if (nucleotide) {
               k_scan_sliced_rowwave<3, 5><<<blocks, 256, 0, hip_stream>>>(planes, filters[q], range.counts[q], dev.row_words, n_positions);
            } else {
               k_scan_sliced_rowwave<5, 22><<<blocks, 256, 0, hip_stream>>>(planes, filters[q], range.counts[q], dev.row_words, n_positions);
            }
         }
      }
      HIP_TRY(hipGetLastError());
      g_last_scan_kernel = "k_scan_sliced_rowwave";
      return SILO_GPU_OK;
   }
   g_last_scan_kernel = q_count == 1 ? "k_scan_sliced" : "k_scan_sliced_batch";
   scanTimingLog().used = 0;
   bool any_derived = false;
   for (const ScanRange& range : caller_ranges) {
      any_derived = any_derived || range.seqstore->layout.has_implicit;
   }
   DerivedPlan plan;
   if (any_derived) {
      planDerived(store, caller_ranges, filters, q_count, plan);
   }
   const int divisor = g_tune_sparse_divisor.load();
   const bool routing = divisor >= 0;
   if (!routing && !any_derived) {
      std::vector<ScanPiece> pieces[N_SCAN_LAYOUTS];
      cutIntoPieces(caller_ranges, q_count, pieces);
      int rc = forkSidePasses(caller_ranges, filters, q_count, nullptr, hip_stream);
      if (rc == SILO_GPU_OK) {
         rc = scanPiecesDense(pieces, any_store, filters, q_count, nullptr, 0, hip_stream);
      }
      const int joined = joinSides(hip_stream);
      return rc != SILO_GPU_OK ? rc : joined;
   }
   const uint32_t capacity = std::max<uint32_t>(4, any_store.row_words / static_cast<uint32_t>(divisor <= 0 ? 16 : divisor));
   SparseScratch* scratch = nullptr;
   const int acquired = acquireSparseScratch(store->device, capacity, plan.table_words, &scratch);
   if (acquired != SILO_GPU_OK) {
      return acquired;
   }
   if (any_derived) {
      bindDerived(plan, *scratch, q_count);
   }
   const std::vector<ScanRange>& ranges = any_derived ? plan.private_ranges : caller_ranges;
   std::vector<ScanPiece> pieces[N_SCAN_LAYOUTS];
   cutIntoPieces(ranges, q_count, pieces);
   const uint32_t stride = scratch->capacity;  // the block may be larger than asked for
   uint32_t* counters = scratch->counters[scratch->set];
   int rc = SILO_GPU_OK;
   {
      // the prepare step: the sectors of every filter that hold a set bit, its cardinality, the private tables zeroed, the
      // other counter set re-armed
      ScanBatchArgs compact{};
      for (uint32_t q = 0; q < q_count; ++q) {
         compact.filters[q] = filters[q];
      }
      k_compact_filter<<<dim3((any_store.row_words + COMPACT_THREADS - 1) / COMPACT_THREADS, q_count), COMPACT_THREADS, 0, hip_stream>>>(
         compact, any_store.row_words, stride, counters, scratch->sector_index, scratch->tables, static_cast<uint32_t>(plan.table_words),
         scratch->counters[scratch->set ^ 1u]
      );
      if (hipGetLastError() != hipSuccess) {
         scratch->set ^= 1u;  // the other set was not re-armed: the next use takes this one again
         releaseSparseScratch(scratch, hip_stream);
         return fail(SILO_GPU_ERR_HIP, "mutations scan: the prepare step could not be launched");
      }
      // the side passes are forked behind the prepare step: the plane scans wait for its counters, and beside a launch that
      // fills the device it takes ten times as long (62 instead of 6 us)
      rc = forkSidePasses(ranges, filters, q_count, any_derived ? &plan : nullptr, hip_stream);
      if (rc == SILO_GPU_OK) {
         rc = scanPiecesDense(pieces, any_store, filters, q_count, routing ? counters : nullptr, capacity, hip_stream);
      }
   }
   // the gather over the sectors of the sparse filters, over the same pieces of the same planes
   for (int layout = 0; routing && layout < N_SCAN_LAYOUTS; ++layout) {
      const std::vector<ScanPiece>& list = pieces[layout];
      for (size_t first_piece = 0; rc == SILO_GPU_OK && first_piece < list.size(); first_piece += SCAN_MAX_RANGES) {
         ScanBatchArgs batch{};
         batch.sparse_sectors = counters;
         batch.sparse_capacity = capacity;
         batch.out_symbols = any_store.n_scan;
         for (uint32_t q = 0; q < q_count; ++q) {
            batch.filters[q] = filters[q];
         }
         enterPieces(batch, list, first_piece, static_cast<uint32_t>(std::min<size_t>(SCAN_MAX_RANGES, list.size() - first_piece)), 0, q_count);
         switch (layout) {
            case SCAN_2_PLANES: rc = launchGatherScan<2, 3, 4, KIND_MAPPED>(batch, scratch->sector_index, stride, any_store.row_words, q_count, hip_stream); break;
            case SCAN_3_PLANES_MAPPED: rc = launchGatherScan<3, 7, 4, KIND_MAPPED>(batch, scratch->sector_index, stride, any_store.row_words, q_count, hip_stream); break;
            case SCAN_FULL_NUCLEOTIDE: rc = launchGatherScan<3, 5, 4, KIND_IDENTITY>(batch, scratch->sector_index, stride, any_store.row_words, q_count, hip_stream); break;
            case SCAN_ONE_HOT_ROWS: rc = launchGatherScan<1, 1, 8, KIND_ROWS>(batch, scratch->sector_index, stride, any_store.row_words, q_count, hip_stream); break;
            default: rc = launchGatherScan<5, 22, 2, KIND_IDENTITY>(batch, scratch->sector_index, stride, any_store.row_words, q_count, hip_stream); break;
         }
      }
   }
   const int joined = joinSides(hip_stream);  // before the scratch is released: side-stream scans read its counters
   if (rc == SILO_GPU_OK && joined == SILO_GPU_OK && any_derived) {
      rc = finishDerived(plan, q_count, hip_stream);
   }
   releaseSparseScratch(scratch, hip_stream);
   return rc != SILO_GPU_OK ? rc : joined;
}

/// finalize(): derive the adaptive code planes of one sequence store (see chooseLayouts) and release its build-time
/// planes — or keep those as they are when re-encoding would not pay (short rows), is switched off
/// (SILO_GPU_TUNE_COMPACT_INDEX < 0) or does not fit next to them.
/// Every position of the store keeps its n_bits identity planes, where they are (short rows, re-encoding switched off, or no room).
int keepBuildPlanes(SeqStoreHost& seqstore) {
   SeqStoreDev& dev = seqstore.dev;
   seqstore.layout.runs.assign(1, SeqStoreHost::Run{0, dev.positions, static_cast<uint8_t>(dev.n_bits), true, false});
   dev.planes = dev.scan;
   dev.row_of = nullptr;
   dev.code_map = nullptr;
   dev.escapes = nullptr;
   dev.escape_first = nullptr;
   seqstore.layout.built = true;
   return SILO_GPU_OK;
}

/// Is this a store finalize re-encodes (5 nucleotide / 22 amino-acid scan symbols, rows of at least one column tile)?
bool reencodes(const silo_gpu_store* store, const SeqStoreDev& dev) {
   const bool nucleotide = dev.n_bits == 3 && dev.n_scan == 5;
   return (nucleotide || (dev.n_bits == 5 && dev.n_scan == 22)) && dev.row_words >= SCAN_THREADS * 4 && dev.positions != 0 && store->sequence_count != 0 &&
          layoutOption(store) >= 0;
}

#define SILO_LAYOUT_TRY(expr)                                                       \
   do {                                                                             \
      const hipError_t status_ = (expr);                                            \
      if (status_ != hipSuccess) {                                                  \
         work.discard();                                                            \
         HIP_TRY(status_);                                                          \
      }                                                                             \
   } while (0)

/// From the totals of the store (seqstore.d_totals): the layout of every position, the tables that describe it and the device
/// arrays of the finished store — the plane rows zeroed when `zero_planes` (an encoder that only sets bits).  *fits = false
/// (nothing allocated) when no position would be re-encoded or the arrays do not fit.
int planLayout(const silo_gpu_store* store, SeqStoreHost& seqstore, SeqStoreHost::LayoutWork& work, bool zero_planes, bool allow_implicit, bool* fits) {
   SeqStoreDev& dev = seqstore.dev;
   const uint32_t positions = dev.positions;
   const size_t n_counters = static_cast<size_t>(positions) * dev.n_scan;
   *fits = false;
   std::vector<uint32_t> totals(n_counters);
   HIP_TRY(hipMemcpy(totals.data(), seqstore.d_totals, n_counters * sizeof(uint32_t), hipMemcpyDeviceToHost));
   std::vector<uint32_t> counts;  // escape keys per (position, symbol)
   // SILO_GPU_TUNE_COMPACT_INDEX 2: code planes only, no one-hot rows; 3: one-hot rows with a row for the most numerous symbol
   // too (the layouts of round 2, for comparisons).  The most numerous symbol of a position is derived (LAYOUT_IMPLICIT) only where
   // the rows without a valid symbol can be counted without a plane: the missing symbol kept as runs.
   const int tuned = layoutOption(store);
   const int one_hot_mode = tuned == 2 ? silo_gpu_layout::ONE_HOT_OFF : (tuned == 3 || !allow_implicit ? silo_gpu_layout::ONE_HOT_ROWS : silo_gpu_layout::ONE_HOT_IMPLICIT);
   silo_gpu_layout::chooseLayouts(
      totals, dev.n_scan, dev.n_bits, positions, static_cast<uint64_t>(dev.row_words) * sizeof(uint64_t), one_hot_mode,
      keyCostOption(store) > 0 ? static_cast<uint64_t>(keyCostOption(store)) : KEY_COST_BYTES, work.code_map, counts,
      launchCostOption(store) == 0 ? silo_gpu_layout::LAUNCH_COST_BYTES : (launchCostOption(store) < 0 ? 0 : static_cast<uint64_t>(launchCostOption(store)) << 10)
   );
   work.row_of.assign(positions + 1, 0);
   work.escape_first.assign(positions + 1, 0);
   work.escape_first_symbol.assign(n_counters + 1, 0);
   bool any_encoded = false;
   for (uint32_t p = 0; p < positions; ++p) {
      const uint8_t* map = work.code_map.data() + static_cast<size_t>(p) * CODE_MAP_STRIDE;
      const uint8_t bits = map[0] & LAYOUT_ROWS_MASK;
      const bool identity = (map[0] & LAYOUT_IDENTITY) != 0;
      const bool one_hot = (map[0] & LAYOUT_ONE_HOT) != 0;
      any_encoded = any_encoded || !identity;
      work.has_implicit = work.has_implicit || (map[0] & LAYOUT_IMPLICIT) != 0;
      work.row_of[p] = static_cast<uint32_t>(work.total_rows);
      work.total_rows += bits;
      for (uint32_t row = 0; row < bits; ++row) {  // a row without a symbol (no valid symbol at the position at all) is empty: any counter of the position
         work.row_target.push_back(one_hot ? p * dev.n_scan + (map[1 + row] != 0xFFu ? map[1 + row] : 0u) : 0xFFFFFFFFu);
      }
      work.escape_first[p] = static_cast<uint32_t>(work.total_escapes);
      for (uint32_t symbol = 0; symbol < dev.n_scan; ++symbol) {
         work.escape_first_symbol[static_cast<size_t>(p) * dev.n_scan + symbol] = static_cast<uint32_t>(work.total_escapes);
         work.total_escapes += counts[static_cast<size_t>(p) * dev.n_scan + symbol];
      }
      const uint8_t run_bits = one_hot ? 0 : bits;
      if (work.runs.empty() || work.runs.back().bits != run_bits || work.runs.back().identity != identity || work.runs.back().one_hot != one_hot) {
         work.runs.push_back(SeqStoreHost::Run{p, p + 1, run_bits, identity, one_hot});
      } else {
         work.runs.back().end = p + 1;
      }
   }
   work.row_of[positions] = static_cast<uint32_t>(work.total_rows);
   work.escape_first[positions] = static_cast<uint32_t>(work.total_escapes);
   work.escape_first_symbol[n_counters] = static_cast<uint32_t>(work.total_escapes);
   work.plane_bytes = static_cast<size_t>(work.total_rows) * dev.row_words * sizeof(uint64_t);
   work.escape_bytes = std::max<uint64_t>(work.total_escapes, 1) * sizeof(uint64_t);
   size_t free_bytes = 0, total_bytes = 0;
   HIP_TRY(hipMemGetInfo(&free_bytes, &total_bytes));
   // the sort of the keys needs as much again as the keys, their slice-major copy as well
   if (!any_encoded || work.total_rows >= (uint64_t{1} << 32) || work.total_escapes >= (uint64_t{1} << 32) ||
       free_bytes < work.plane_bytes + 4 * work.escape_bytes + (size_t{1} << 30)) {
      work = SeqStoreHost::LayoutWork{};
      return SILO_GPU_OK;
   }
   SILO_LAYOUT_TRY(hipMalloc(&work.d_code_map, work.code_map.size()));
   SILO_LAYOUT_TRY(hipMalloc(&work.d_cursor, n_counters * sizeof(uint32_t)));
   SILO_LAYOUT_TRY(hipMalloc(&work.d_planes, std::max<size_t>(work.plane_bytes, 256)));
   SILO_LAYOUT_TRY(hipMalloc(&work.d_escapes, work.escape_bytes));
   SILO_LAYOUT_TRY(hipMalloc(&work.d_first, work.escape_first_symbol.size() * sizeof(uint32_t)));
   SILO_LAYOUT_TRY(hipMalloc(&work.d_row_of, work.row_of.size() * sizeof(uint32_t)));
   SILO_LAYOUT_TRY(hipMalloc(&work.d_escape_first, work.escape_first.size() * sizeof(uint32_t)));
   SILO_LAYOUT_TRY(hipMalloc(&work.d_row_target, std::max<size_t>(work.row_target.size(), 1) * sizeof(uint32_t)));
   SILO_LAYOUT_TRY(hipMemcpy(work.d_code_map, work.code_map.data(), work.code_map.size(), hipMemcpyHostToDevice));
   SILO_LAYOUT_TRY(hipMemcpy(work.d_first, work.escape_first_symbol.data(), work.escape_first_symbol.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
   SILO_LAYOUT_TRY(hipMemcpy(work.d_row_of, work.row_of.data(), work.row_of.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
   SILO_LAYOUT_TRY(hipMemcpy(work.d_row_target, work.row_target.data(), work.row_target.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
   SILO_LAYOUT_TRY(hipMemcpy(work.d_escape_first, work.escape_first.data(), work.escape_first.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
   SILO_LAYOUT_TRY(hipMemset(work.d_cursor, 0, n_counters * sizeof(uint32_t)));
   if (zero_planes) {  // the encoding pass of a two-pass build only sets bits and fills key slots: a slot it misses must not look like a key
      SILO_LAYOUT_TRY(hipMemset(work.d_planes, 0, std::max<size_t>(work.plane_bytes, 256)));
      SILO_LAYOUT_TRY(hipMemset(work.d_escapes, 0xFF, work.escape_bytes));
   }
   SILO_LAYOUT_TRY(hipStreamSynchronize(nullptr));  // the fills are only enqueued (null stream)
   *fits = true;
   return SILO_GPU_OK;
}

/// The planned layout becomes the store: the keys are sorted (and copied slice-major for the scan's escape pass), the
/// encoders' tables and any build-time planes are released, the device description switches to the adaptive planes.
int finishLayout(silo_gpu_store* store, SeqStoreHost& seqstore, SeqStoreHost::LayoutWork& work) {
   SeqStoreDev& dev = seqstore.dev;
   SeqStoreHost::Layout& layout = seqstore.layout;
   const uint32_t positions = dev.positions;
   if (const int rc = silo_gpu_internal_sort_keys(work.d_escapes, work.total_escapes); rc != SILO_GPU_OK) {  // ascending: (position, symbol, sequence)
      work.discard();
      return rc;
   }
   // the slice-major copy of the keys for the scan's escape pass, and where each position's keys begin in every slice
   uint64_t* d_escapes_sliced = nullptr;
   uint32_t* d_slice_first = nullptr;
   std::vector<uint32_t> slice_first;
   const uint32_t n_slices = (store->sequence_count + (1u << ESCAPE_SLICE_SHIFT) - 1) >> ESCAPE_SLICE_SHIFT;
   if (work.total_escapes > 0 && n_slices <= ESCAPE_MAX_SLICES) {
      const size_t n_entries = static_cast<size_t>(n_slices) * (positions + 1);
      const auto discardSliced = [&]() {
         (void)hipFree(d_escapes_sliced);
         (void)hipFree(d_slice_first);
      };
      hipError_t status = hipMalloc(&d_escapes_sliced, work.escape_bytes + 16);  // (the scan loads the keys in pairs: one key of slack)
      status = status != hipSuccess ? status : hipMalloc(&d_slice_first, n_entries * sizeof(uint32_t));
      status = status != hipSuccess ? status : hipMemcpy(d_escapes_sliced, work.d_escapes, work.total_escapes * sizeof(uint64_t), hipMemcpyDeviceToDevice);
      if (status != hipSuccess) {
         discardSliced();
         SILO_LAYOUT_TRY(status);
      }
      if (const int rc = silo_gpu_internal_sort_keys_by_bits(d_escapes_sliced, work.total_escapes, ESCAPE_SLICE_SHIFT, ESCAPE_SLICE_SHIFT + ESCAPE_SLICE_BITS); rc != SILO_GPU_OK) {
         discardSliced();
         work.discard();
         return rc;
      }
      k_slice_index<<<static_cast<uint32_t>((n_entries + 255) / 256), 256>>>(
         d_escapes_sliced, static_cast<uint32_t>(work.total_escapes), ESCAPE_SLICE_SHIFT, n_slices, positions, d_slice_first
      );
      k_recode_sliced_keys<<<static_cast<uint32_t>((work.total_escapes + 255) / 256), 256>>>(d_escapes_sliced, static_cast<uint32_t>(work.total_escapes), dev.n_scan);
      slice_first.resize(n_entries);
      status = hipGetLastError();
      status = status != hipSuccess ? status : hipMemcpy(slice_first.data(), d_slice_first, n_entries * sizeof(uint32_t), hipMemcpyDeviceToHost);
      if (status != hipSuccess) {
         discardSliced();
         SILO_LAYOUT_TRY(status);
      }
   }
   (void)hipFree(work.d_first);
   (void)hipFree(work.d_cursor);
   work.d_first = nullptr;
   work.d_cursor = nullptr;
   if (dev.scan != nullptr) {  // the adaptive planes take over; the build-time planes go
      const size_t build_bytes = static_cast<size_t>(positions) * dev.n_bits * dev.row_words * sizeof(uint64_t);
      (void)hipFree(dev.scan);
      dev.scan = nullptr;
      store->device_bytes -= build_bytes;
   }
   layout.planes = work.d_planes;
   layout.d_row_of = work.d_row_of;
   layout.d_row_target = work.d_row_target;
   layout.d_code_map = work.d_code_map;
   layout.d_escapes = work.d_escapes;
   layout.d_escapes_sliced = d_escapes_sliced;
   layout.d_slice_first = d_slice_first;
   layout.slice_shift = ESCAPE_SLICE_SHIFT;
   layout.n_slices = d_escapes_sliced != nullptr ? n_slices : 0;
   layout.slice_first = std::move(slice_first);
   layout.d_escape_first = work.d_escape_first;
   layout.row_of = std::move(work.row_of);
   layout.code_map = std::move(work.code_map);
   layout.escape_first = std::move(work.escape_first);
   layout.escape_first_symbol = std::move(work.escape_first_symbol);
   layout.runs = std::move(work.runs);
   layout.has_implicit = work.has_implicit;
   layout.device_bytes = work.plane_bytes + work.escape_bytes * (d_escapes_sliced != nullptr ? 2 : 1) + static_cast<size_t>(positions) * (CODE_MAP_STRIDE + 8) +
                         work.total_rows * sizeof(uint32_t);
   store->device_bytes += layout.device_bytes;
   dev.planes = layout.planes;
   dev.row_of = layout.d_row_of;
   dev.code_map = layout.d_code_map;
   dev.escapes = layout.d_escapes;
   dev.escape_first = layout.d_escape_first;
   layout.built = true;
   work = SeqStoreHost::LayoutWork{};  // everything it owned is the store's now
   return SILO_GPU_OK;
}

/// +1 where a run of the missing symbol starts, -1 where it ends: summed along the positions, the rows with the missing symbol.
__global__ void k_runs_diff_all(const uint64_t* __restrict__ run_keys, const uint32_t* __restrict__ run_ends, uint32_t n_runs, uint32_t* __restrict__ diff) {
   const uint32_t run = blockIdx.x * blockDim.x + threadIdx.x;
   if (run < n_runs) {
      atomicAdd(&diff[static_cast<uint32_t>(run_keys[run])], 1u);
      atomicAdd(&diff[run_ends[run]], 0xFFFFFFFFu);
   }
}

/// Does every row of the store have a symbol at every position — a valid one (the totals), the missing one (its runs) or a
/// sparsely stored one (the sorted keys)?  Rows that never received a sequence, or an import whose bitmaps leave rows out,
/// do not; such a store derives nothing (the derived symbol would take them in).
int everyRowHasASymbol(const silo_gpu_store* store, const SeqStoreHost& seqstore, bool* complete) {
   const SeqStoreDev& dev = seqstore.dev;
   const uint32_t positions = dev.positions;
   *complete = false;
   if (seqstore.d_totals == nullptr || !seqstore.totals_ready || dev.kind[dev.missing_symbol] != PLANE_RUNS) {
      return SILO_GPU_OK;
   }
   std::vector<uint32_t> totals(static_cast<size_t>(positions) * dev.n_scan);
   HIP_TRY(hipMemcpy(totals.data(), seqstore.d_totals, totals.size() * sizeof(uint32_t), hipMemcpyDeviceToHost));
   std::vector<uint32_t> diff(positions + 1, 0);
   if (dev.n_missing_runs != 0) {
      uint32_t* d_diff = nullptr;
      HIP_TRY(hipMalloc(&d_diff, diff.size() * sizeof(uint32_t)));
      hipError_t status = hipMemset(d_diff, 0, diff.size() * sizeof(uint32_t));
      if (status == hipSuccess) {
         k_runs_diff_all<<<(dev.n_missing_runs + 255) / 256, 256>>>(dev.missing_run_keys, dev.missing_run_ends, dev.n_missing_runs, d_diff);
         status = hipGetLastError();
      }
      status = status != hipSuccess ? status : hipMemcpy(diff.data(), d_diff, diff.size() * sizeof(uint32_t), hipMemcpyDeviceToHost);
      (void)hipFree(d_diff);
      HIP_TRY(status);
   }
   std::vector<uint32_t> sparse(positions, 0);
   for (const uint64_t key : seqstore.sparse_sorted) {
      const uint64_t position = key >> 37;
      if (position < positions) {
         sparse[position] += 1;
      }
   }
   uint32_t missing = 0;
   for (uint32_t p = 0; p < positions; ++p) {
      missing += diff[p];
      uint64_t covered = static_cast<uint64_t>(missing) + sparse[p];
      for (uint32_t symbol = 0; symbol < dev.n_scan; ++symbol) {
         covered += totals[static_cast<size_t>(p) * dev.n_scan + symbol];
      }
      if (covered != store->sequence_count) {
         return SILO_GPU_OK;
      }
   }
   *complete = true;
   return SILO_GPU_OK;
}

/// finalize(): derive the adaptive planes of one sequence store from its build-time planes and release those — or keep them as
/// they are when re-encoding would not pay (short rows), is switched off (SILO_GPU_TUNE_COMPACT_INDEX < 0) or does not fit
/// next to them.  A store built in two passes has been encoded already: only the keys remain to be put in order.
int buildLayout(silo_gpu_store* store, SeqStoreHost& seqstore) {
   SeqStoreDev& dev = seqstore.dev;
   SeqStoreHost::Layout& layout = seqstore.layout;
   const uint32_t positions = dev.positions;
   if (dev.build_mode == BUILD_ENCODE) {
      HIP_TRY(hipDeviceSynchronize());
      dev.build_mode = BUILD_PLANES;
      {  // every (position, symbol) must have received exactly the keys the first pass counted for it
         SeqStoreHost::LayoutWork& work = *seqstore.work;
         const uint32_t n_counters = positions * dev.n_scan;
         uint32_t* d_mismatch = nullptr;
         uint32_t mismatch = 0;
         HIP_TRY(hipMalloc(&d_mismatch, sizeof(uint32_t)));
         hipError_t status = hipMemset(d_mismatch, 0, sizeof(uint32_t));
         if (status == hipSuccess && n_counters != 0) {
            k_check_cursors<<<(n_counters + 255) / 256, 256>>>(work.d_first, work.d_cursor, n_counters, d_mismatch);
            status = hipGetLastError();
         }
         status = status != hipSuccess ? status : hipMemcpy(&mismatch, d_mismatch, sizeof(uint32_t), hipMemcpyDeviceToHost);
         (void)hipFree(d_mismatch);
         if (status != hipSuccess || mismatch != 0) {
            work.discard();
            seqstore.work.reset();
            HIP_TRY(status);
            return fail(SILO_GPU_ERR_INVALID_ARGUMENT, "two-pass build: the second pass did not bring the rows the first pass counted (escape keys of " + std::to_string(mismatch) + " (position, symbol) cells differ)");
         }
      }
      if (seqstore.work->has_implicit) {
         bool complete = false;
         if (const int rc = everyRowHasASymbol(store, seqstore, &complete); rc != SILO_GPU_OK) {
            return rc;
         }
         if (!complete) {
            seqstore.work->discard();
            seqstore.work.reset();
            return fail(SILO_GPU_ERR_INVALID_ARGUMENT, "two-pass build: some row of the store has no symbol at some position (every row has to be filled in both passes)");
         }
      }
      return finishLayout(store, seqstore, *seqstore.work);
   }
   if (!reencodes(store, dev)) {
      return keepBuildPlanes(seqstore);
   }
   // the unfiltered totals decide the codes (and are what a full filter adds later on)
   const size_t n_counters = static_cast<size_t>(positions) * dev.n_scan;
   if (seqstore.d_totals == nullptr) {
      HIP_TRY(hipMalloc(&seqstore.d_totals, n_counters * sizeof(uint32_t)));
   }
   if (!seqstore.totals_ready) {
      HIP_TRY(hipMemsetAsync(seqstore.d_totals, 0, n_counters * sizeof(uint32_t), nullptr));
      ScanRange all{&seqstore, 0, positions, {}};
      all.counts[0] = seqstore.d_totals;
      const uint64_t* ones = store->d_ones;
      layout.runs.clear();  // scan the build-time planes
      const int rc = scanRanges(store, {all}, &ones, 1, nullptr);
      if (rc != SILO_GPU_OK) {
         return rc;
      }
      HIP_TRY(hipStreamSynchronize(nullptr));
      seqstore.totals_ready = true;
   }
   SeqStoreHost::LayoutWork work;
   bool fits = false;
   bool complete = false;  // only a store whose every row has a symbol at every position may derive one as "the rest"
   if (dev.kind[dev.missing_symbol] == PLANE_RUNS) {
      if (const int rc = everyRowHasASymbol(store, seqstore, &complete); rc != SILO_GPU_OK) {
         return rc;
      }
   }
   if (const int rc = planLayout(store, seqstore, work, false, complete, &fits); rc != SILO_GPU_OK) {
      return rc;
   }
   if (!fits) {
      return keepBuildPlanes(seqstore);
   }
   {
      const dim3 grid((dev.row_words + 255) / 256, positions);
      if (dev.n_bits == 3) {
         k_encode_adaptive<3><<<grid, 256>>>(dev.scan, dev.row_words, dev.n_scan, work.d_code_map, work.d_row_of, work.d_first, work.d_cursor, work.d_planes, work.d_escapes);
      } else {
         k_encode_adaptive<5><<<grid, 256>>>(dev.scan, dev.row_words, dev.n_scan, work.d_code_map, work.d_row_of, work.d_first, work.d_cursor, work.d_planes, work.d_escapes);
      }
      SILO_LAYOUT_TRY(hipGetLastError());
      SILO_LAYOUT_TRY(hipDeviceSynchronize());
   }
   return finishLayout(store, seqstore, work);
}
#undef SILO_LAYOUT_TRY

}  // namespace

extern "C" {

int silo_gpu_mutations_scan_ranges(
   const silo_gpu_store* store, const silo_gpu_scan_range* ranges, uint32_t n_ranges, const uint64_t* const* filters_dev, uint32_t n_filters,
   uint32_t* const* counts_out_dev, void* stream
) {
   if (store == nullptr || (n_ranges != 0 && ranges == nullptr) || (n_filters != 0 && filters_dev == nullptr) ||
       (n_ranges != 0 && n_filters != 0 && counts_out_dev == nullptr)) {
      return fail(SILO_GPU_ERR_INVALID_ARGUMENT, "silo_gpu_mutations_scan_ranges: bad arguments");
   }
   for (uint32_t q = 0; q < n_filters; ++q) {
      if (filters_dev[q] == nullptr) {
         return fail(SILO_GPU_ERR_INVALID_ARGUMENT, "silo_gpu_mutations_scan_ranges: null filter");
      }
   }
   for (uint32_t r = 0; r < n_ranges; ++r) {
      if (ranges[r].seqstore_id >= store->seqstores.size()) {
         return fail(SILO_GPU_ERR_INVALID_ARGUMENT, "silo_gpu_mutations_scan_ranges: no such sequence store");
      }
      const SeqStoreDev& dev = store->seqstores[ranges[r].seqstore_id].dev;
      if (ranges[r].pos_begin > ranges[r].pos_end || ranges[r].pos_end > dev.positions) {
         return fail(SILO_GPU_ERR_INVALID_ARGUMENT, "position range out of bounds");
      }
      for (uint32_t q = 0; q < n_filters; ++q) {
         if (counts_out_dev[static_cast<size_t>(r) * n_filters + q] == nullptr) {
            return fail(SILO_GPU_ERR_INVALID_ARGUMENT, "silo_gpu_mutations_scan_ranges: null counts buffer");
         }
      }
   }
   if (n_ranges == 0 || n_filters == 0) {
      return SILO_GPU_OK;
   }
   HIP_TRY(hipSetDevice(store->device));
   auto hip_stream = static_cast<hipStream_t>(stream);
   for (uint32_t first = 0; first < n_filters; first += SILO_GPU_MAX_SCAN_BATCH) {
      const uint32_t q_count = std::min<uint32_t>(SILO_GPU_MAX_SCAN_BATCH, n_filters - first);
      // ranges of one layout (3 code planes / 5 code planes) share launches
      for (const uint32_t n_bits : {3u, 5u}) {
         std::vector<ScanRange> group;
         for (uint32_t r = 0; r < n_ranges; ++r) {
            const SeqStoreDev& dev = store->seqstores[ranges[r].seqstore_id].dev;
            if (ranges[r].pos_begin == ranges[r].pos_end || dev.n_scan == 0 || (dev.n_bits == 3 ? 3u : 5u) != n_bits) {
               continue;
            }
            const SeqStoreHost& seqstore = store->seqstores[ranges[r].seqstore_id];
            if (dev.planes == nullptr) {
               return fail(SILO_GPU_ERR_INVALID_ARGUMENT, "silo_gpu_mutations_scan_ranges: the sequence store holds no sequences yet");
            }
            ScanRange range{&seqstore, ranges[r].pos_begin, ranges[r].pos_end, {}};
            for (uint32_t q = 0; q < q_count; ++q) {
               range.counts[q] = counts_out_dev[static_cast<size_t>(r) * n_filters + first + q];
            }
            group.push_back(range);
         }
         if (!group.empty()) {
            const int rc = scanRanges(store, group, filters_dev + first, q_count, hip_stream);
            if (rc != SILO_GPU_OK) {
               return rc;
            }
         }
      }
   }
   return SILO_GPU_OK;
}

int silo_gpu_scan_timings(silo_gpu_scan_timing* out, uint32_t capacity, uint32_t* n_out) {
   if (n_out == nullptr || (capacity != 0 && out == nullptr)) {
      return fail(SILO_GPU_ERR_INVALID_ARGUMENT, "silo_gpu_scan_timings: bad arguments");
   }
   ScanTimingLog& log = scanTimingLog();
   *n_out = static_cast<uint32_t>(log.used);
   for (size_t k = 0; k < log.used && k < capacity; ++k) {
      ScanLaunchTiming& launch = log.launches[k];
      HIP_TRY(hipEventSynchronize(launch.stop));
      HIP_TRY(hipEventElapsedTime(&launch.entry.ms, launch.start, launch.stop));
      out[k] = launch.entry;
   }
   return SILO_GPU_OK;
}

int silo_gpu_mutations_scan_batch(
   const silo_gpu_store* store, uint32_t seqstore_id, const uint64_t* const* filters_dev, uint32_t n_filters, uint32_t pos_begin,
   uint32_t pos_end, uint32_t* const* counts_out_dev, void* stream
) {
   if (store == nullptr || seqstore_id >= store->seqstores.size() || filters_dev == nullptr || counts_out_dev == nullptr) {
      return fail(SILO_GPU_ERR_INVALID_ARGUMENT, "silo_gpu_mutations_scan_batch: bad arguments");
   }
   const silo_gpu_scan_range range{seqstore_id, pos_begin, pos_end};
   return silo_gpu_mutations_scan_ranges(store, &range, 1, filters_dev, n_filters, counts_out_dev, stream);
}

uint32_t silo_gpu_store_scan_planes(const silo_gpu_store* store, uint32_t seqstore_id) {
   if (store == nullptr || seqstore_id >= store->seqstores.size()) {
      return 0;
   }
   const SeqStoreHost& seqstore = store->seqstores[seqstore_id];
   uint64_t positions_with[8] = {0, 0, 0, 0, 0, 0, 0, 0};  // by number of plane rows
   for (const SeqStoreHost::Run& run : seqstore.layout.runs) {
      for (uint32_t p = run.begin; run.one_hot && p < run.end; ++p) {
         positions_with[(seqstore.layout.row_of[p + 1] - seqstore.layout.row_of[p]) & 7u] += 1;
      }
      positions_with[run.bits & 7u] += run.one_hot ? 0 : run.end - run.begin;
   }
   uint32_t most_common = seqstore.dev.n_bits;
   uint64_t most = 0;
   for (uint32_t bits = 0; bits < 8; ++bits) {  // (0: positions whose only frequent symbol is derived)
      if (positions_with[bits] > most) {
         most = positions_with[bits];
         most_common = bits;
      }
   }
   return most_common;
}

uint64_t silo_gpu_store_scan_rows(const silo_gpu_store* store, uint32_t seqstore_id, uint32_t pos_begin, uint32_t pos_end) {
   if (store == nullptr || seqstore_id >= store->seqstores.size()) {
      return 0;
   }
   const SeqStoreHost& seqstore = store->seqstores[seqstore_id];
   pos_end = std::min(pos_end, seqstore.dev.positions);
   if (pos_begin >= pos_end) {
      return 0;
   }
   if (seqstore.layout.row_of.empty()) {
      return static_cast<uint64_t>(pos_end - pos_begin) * seqstore.dev.n_bits;
   }
   return seqstore.layout.row_of[pos_end] - seqstore.layout.row_of[pos_begin];
}

uint64_t silo_gpu_store_scan_runs(const silo_gpu_store* store, uint32_t seqstore_id) {
   if (store == nullptr || seqstore_id >= store->seqstores.size() || !store->seqstores[seqstore_id].layout.has_implicit) {
      return 0;
   }
   return store->seqstores[seqstore_id].dev.n_missing_runs;
}

uint64_t silo_gpu_store_scan_sparse_keys(const silo_gpu_store* store, uint32_t seqstore_id) {
   if (store == nullptr || seqstore_id >= store->seqstores.size() || !store->seqstores[seqstore_id].layout.has_implicit) {
      return 0;
   }
   return store->seqstores[seqstore_id].sparse_sorted.size();
}

uint64_t silo_gpu_store_scan_escapes(const silo_gpu_store* store, uint32_t seqstore_id) {
   if (store == nullptr || seqstore_id >= store->seqstores.size() || store->seqstores[seqstore_id].layout.escape_first.empty()) {
      return 0;
   }
   return store->seqstores[seqstore_id].layout.escape_first.back();
}

int silo_gpu_memset_async(void* dev_ptr, int value, size_t bytes, void* stream) {
   HIP_TRY(hipMemsetAsync(dev_ptr, value, bytes, static_cast<hipStream_t>(stream)));
   return SILO_GPU_OK;
}

int silo_gpu_memcpy_d2h(void* dst_host, const void* src_dev, size_t bytes, void* stream) {
   // Results (count shards, the counts[P][S] table) are small: bounce them through a per-thread pinned
   // buffer so the copy is one DMA instead of the runtime's staged pageable path.
   constexpr size_t PINNED_BYTES = 4u << 20;
   struct Pinned {  // never freed: thread exit may come after the HIP runtime has shut down
      void* ptr = nullptr;
      bool tried = false;
   };
   thread_local Pinned pinned;
   if (bytes <= PINNED_BYTES && !pinned.tried) {
      pinned.tried = true;
      if (hipHostMalloc(&pinned.ptr, PINNED_BYTES, hipHostMallocDefault) != hipSuccess) {
         (void)hipGetLastError();
         pinned.ptr = nullptr;
      }
   }
   auto hip_stream = static_cast<hipStream_t>(stream);
   if (bytes <= PINNED_BYTES && pinned.ptr != nullptr) {
      HIP_TRY(hipMemcpyAsync(pinned.ptr, src_dev, bytes, hipMemcpyDeviceToHost, hip_stream));
      HIP_TRY(hipStreamSynchronize(hip_stream));
      memcpy(dst_host, pinned.ptr, bytes);
      return SILO_GPU_OK;
   }
   HIP_TRY(hipMemcpyAsync(dst_host, src_dev, bytes, hipMemcpyDeviceToHost, hip_stream));
   HIP_TRY(hipStreamSynchronize(hip_stream));
   return SILO_GPU_OK;
}

int silo_gpu_host_alloc(size_t bytes, void** out_host) {
   if (out_host == nullptr || bytes == 0) {
      return fail(SILO_GPU_ERR_INVALID_ARGUMENT, "silo_gpu_host_alloc: bad arguments");
   }
   HIP_TRY(hipHostMalloc(out_host, bytes, hipHostMallocDefault));
   return SILO_GPU_OK;
}

void silo_gpu_host_free(void* host) {
   if (host != nullptr) {
      (void)hipHostFree(host);
   }
}

int silo_gpu_memcpy_d2h_async(void* dst_pinned_host, const void* src_dev, size_t bytes, void* stream) {
   if (dst_pinned_host == nullptr || src_dev == nullptr) {
      return fail(SILO_GPU_ERR_INVALID_ARGUMENT, "silo_gpu_memcpy_d2h_async: null pointer");
   }
   HIP_TRY(hipMemcpyAsync(dst_pinned_host, src_dev, bytes, hipMemcpyDeviceToHost, static_cast<hipStream_t>(stream)));
   return SILO_GPU_OK;
}

int silo_gpu_memcpy_h2d(void* dst_dev, const void* src_host, size_t bytes, void* stream) {
   HIP_TRY(hipMemcpyAsync(dst_dev, src_host, bytes, hipMemcpyHostToDevice, static_cast<hipStream_t>(stream)));
   HIP_TRY(hipStreamSynchronize(static_cast<hipStream_t>(stream)));
   return SILO_GPU_OK;
}

int silo_gpu_event_create(void** out_event) {
   if (out_event == nullptr) {
      return fail(SILO_GPU_ERR_INVALID_ARGUMENT, "silo_gpu_event_create: null out pointer");
   }
   hipEvent_t event = nullptr;
   HIP_TRY(hipEventCreate(&event));
   *out_event = event;
   return SILO_GPU_OK;
}

int silo_gpu_event_synchronize(void* event) {
   HIP_TRY(hipEventSynchronize(static_cast<hipEvent_t>(event)));
   return SILO_GPU_OK;
}

int silo_gpu_event_record(void* event, void* stream) {
   HIP_TRY(hipEventRecord(static_cast<hipEvent_t>(event), static_cast<hipStream_t>(stream)));
   return SILO_GPU_OK;
}

int silo_gpu_event_elapsed_ms(void* start_event, void* stop_event, float* out_ms) {
   HIP_TRY(hipEventSynchronize(static_cast<hipEvent_t>(stop_event)));
   HIP_TRY(hipEventElapsedTime(out_ms, static_cast<hipEvent_t>(start_event), static_cast<hipEvent_t>(stop_event)));
   return SILO_GPU_OK;
}

void silo_gpu_event_destroy(void* event) {
   (void)hipEventDestroy(static_cast<hipEvent_t>(event));
}

int silo_gpu_set_device(int device) {
   HIP_TRY(hipSetDevice(device));
   return SILO_GPU_OK;
}

int silo_gpu_stream_create(void** out_stream) {
   if (out_stream == nullptr) {
      return fail(SILO_GPU_ERR_INVALID_ARGUMENT, "silo_gpu_stream_create: null out pointer");
   }
   hipStream_t stream = nullptr;
   HIP_TRY(hipStreamCreateWithFlags(&stream, hipStreamNonBlocking));
   *out_stream = stream;
   return SILO_GPU_OK;
}

void silo_gpu_stream_destroy(void* stream) {
   if (stream != nullptr) {
      (void)hipStreamDestroy(static_cast<hipStream_t>(stream));
   }
}

int silo_gpu_stream_synchronize(void* stream) {
   HIP_TRY(hipStreamSynchronize(static_cast<hipStream_t>(stream)));
   return SILO_GPU_OK;
}

int silo_gpu_bitset_alloc(const silo_gpu_store* store, uint64_t** out_dev) {
   if (store == nullptr || out_dev == nullptr) {
      return fail(SILO_GPU_ERR_INVALID_ARGUMENT, "silo_gpu_bitset_alloc: bad arguments");
   }
   HIP_TRY(hipSetDevice(store->device));
   const size_t bytes = static_cast<size_t>(store->row_words) * sizeof(uint64_t);
   HIP_TRY(hipMalloc(out_dev, bytes));
   HIP_TRY(hipMemset(*out_dev, 0, bytes));
   HIP_TRY(hipStreamSynchronize(nullptr));  // the fill is enqueued on the null stream; the caller's stream would not wait for it
   return SILO_GPU_OK;
}

int silo_gpu_bitset_upload(const silo_gpu_store* store, uint64_t* dst_dev, const uint64_t* src_host, size_t n_words, void* stream) {
   if (store == nullptr || dst_dev == nullptr || src_host == nullptr || n_words > store->row_words) {
      return fail(SILO_GPU_ERR_INVALID_ARGUMENT, "silo_gpu_bitset_upload: bad arguments");
   }
   auto hip_stream = static_cast<hipStream_t>(stream);
   HIP_TRY(hipMemsetAsync(dst_dev, 0, static_cast<size_t>(store->row_words) * sizeof(uint64_t), hip_stream));
   HIP_TRY(hipMemcpyAsync(dst_dev, src_host, n_words * sizeof(uint64_t), hipMemcpyHostToDevice, hip_stream));
   HIP_TRY(hipStreamSynchronize(hip_stream));
   return SILO_GPU_OK;
}

int silo_gpu_bitset_download(const silo_gpu_store* store, uint64_t* dst_host, const uint64_t* src_dev, size_t n_words, void* stream) {
   if (store == nullptr || dst_host == nullptr || src_dev == nullptr || n_words > store->row_words) {
      return fail(SILO_GPU_ERR_INVALID_ARGUMENT, "silo_gpu_bitset_download: bad arguments");
   }
   auto hip_stream = static_cast<hipStream_t>(stream);
   HIP_TRY(hipMemcpyAsync(dst_host, src_dev, n_words * sizeof(uint64_t), hipMemcpyDeviceToHost, hip_stream));
   HIP_TRY(hipStreamSynchronize(hip_stream));
   return SILO_GPU_OK;
}

int silo_gpu_bitset_from_lineages(const silo_gpu_store* store, uint64_t* dst_dev, const uint8_t* membership_by_lineage, uint32_t n_lineages, void* stream) {
   if (store == nullptr || dst_dev == nullptr || membership_by_lineage == nullptr) {
      return fail(SILO_GPU_ERR_INVALID_ARGUMENT, "silo_gpu_bitset_from_lineages: bad arguments");
   }
   if (store->d_lineage == nullptr || n_lineages != store->n_lineages) {
      return fail(SILO_GPU_ERR_INVALID_ARGUMENT, "store holds no synthetic lineage assignment of that size");
   }
   auto hip_stream = static_cast<hipStream_t>(stream);
   uint8_t* d_membership = nullptr;
   HIP_TRY(hipMalloc(&d_membership, n_lineages));
   hipError_t err = hipMemcpyAsync(d_membership, membership_by_lineage, n_lineages, hipMemcpyHostToDevice, hip_stream);
   if (err == hipSuccess) {
      const uint32_t threads = store->row_words * 64u;
      k_bitset_from_lineages<<<(threads + 255) / 256, 256, 0, hip_stream>>>(
         store->d_lineage, d_membership, store->sequence_count, store->row_words, dst_dev
      );
      err = hipStreamSynchronize(hip_stream);
   }
   (void)hipFree(d_membership);
   if (err != hipSuccess) {
      return fail(SILO_GPU_ERR_HIP, std::string("k_bitset_from_lineages: ") + hipGetErrorString(err));
   }
   return SILO_GPU_OK;
}

int silo_gpu_upload_u32(const uint32_t* src_host, size_t n, uint32_t** out_dev) {
   if (out_dev == nullptr || (src_host == nullptr && n > 0)) {
      return fail(SILO_GPU_ERR_INVALID_ARGUMENT, "silo_gpu_upload_u32: bad arguments");
   }
   uint32_t* ptr = nullptr;
   HIP_TRY(hipMalloc(&ptr, std::max<size_t>(n, 1) * sizeof(uint32_t)));
   hipError_t err = hipMemcpy(ptr, src_host, n * sizeof(uint32_t), hipMemcpyHostToDevice);
   if (err != hipSuccess) {
      (void)hipFree(ptr);
      return fail(SILO_GPU_ERR_HIP, std::string("silo_gpu_upload_u32: ") + hipGetErrorString(err));
   }
   *out_dev = ptr;
   return SILO_GPU_OK;
}

int silo_gpu_bitset_from_value_ids(const silo_gpu_store* store, uint64_t* dst_dev, const uint32_t* value_ids_dev, const uint8_t* membership_by_value, uint32_t n_values, void* stream) {
   if (store == nullptr || dst_dev == nullptr || value_ids_dev == nullptr || membership_by_value == nullptr || n_values == 0) {
      return fail(SILO_GPU_ERR_INVALID_ARGUMENT, "silo_gpu_bitset_from_value_ids: bad arguments");
   }
   auto hip_stream = static_cast<hipStream_t>(stream);
   uint8_t* d_membership = nullptr;
   HIP_TRY(hipMalloc(&d_membership, n_values));
   hipError_t err = hipMemcpyAsync(d_membership, membership_by_value, n_values, hipMemcpyHostToDevice, hip_stream);
   if (err == hipSuccess) {
      const uint32_t threads = store->row_words * 64u;
      k_bitset_from_value_ids<<<(threads + 255) / 256, 256, 0, hip_stream>>>(
         value_ids_dev, d_membership, n_values, store->sequence_count, store->row_words, dst_dev
      );
      err = hipStreamSynchronize(hip_stream);
   }
   (void)hipFree(d_membership);
   if (err != hipSuccess) {
      return fail(SILO_GPU_ERR_HIP, std::string("k_bitset_from_value_ids: ") + hipGetErrorString(err));
   }
   return SILO_GPU_OK;
}

const uint64_t* silo_gpu_store_plane(const silo_gpu_store* store, uint32_t seqstore_id, uint32_t position, uint32_t symbol) {
   if (store == nullptr || seqstore_id >= store->seqstores.size()) {
      return nullptr;
   }
   const SeqStoreDev& dev = store->seqstores[seqstore_id].dev;
   if (position >= dev.positions || symbol >= dev.n_symbols) {
      return nullptr;
   }
   return planePtr(dev, position, symbol);
}

int silo_gpu_store_sparse_plane(const silo_gpu_store* store, uint32_t seqstore_id, uint32_t position, uint32_t symbol, uint64_t* dst_dev, void* stream) {
   if (store == nullptr || seqstore_id >= store->seqstores.size() || dst_dev == nullptr) {
      return fail(SILO_GPU_ERR_INVALID_ARGUMENT, "silo_gpu_store_sparse_plane: bad arguments");
   }
   const SeqStoreHost& seqstore = store->seqstores[seqstore_id];
   if (position >= seqstore.dev.positions || symbol >= seqstore.dev.n_symbols) {
      return fail(SILO_GPU_ERR_INVALID_ARGUMENT, "position or symbol out of range");
   }
   if (!seqstore.finalized) {
      return fail(SILO_GPU_ERR_INVALID_ARGUMENT, "store not finalized");
   }
   HIP_TRY(hipSetDevice(store->device));
   auto hip_stream = static_cast<hipStream_t>(stream);
   if (seqstore.dev.kind[symbol] == PLANE_SCAN) {  // a valid mutation symbol: decode its one-hot plane from the position's code planes
      k_decode_plane<<<(store->row_words + 255) / 256, 256, 0, hip_stream>>>(seqstore.dev, position, symbol, store->d_ones, dst_dev);
      HIP_TRY(hipGetLastError());
      const uint8_t* map = seqstore.layout.code_map.empty() ? nullptr : seqstore.layout.code_map.data() + static_cast<size_t>(position) * CODE_MAP_STRIDE;
      if (map != nullptr && (map[0] & LAYOUT_IMPLICIT) != 0 && map[IMPLICIT_SLOT] == seqstore.dev.index[symbol]) {
         // the position's derived symbol: "no other symbol and not missing" (the reference rebuilds its deleted bitmap the same way,
         // nucleotide_symbol_equals.cpp:158-180) — the kernel took the stored rows away; now the keys, the runs, the ambiguity codes
         const uint32_t key_begin = seqstore.layout.escape_first[position];
         const uint32_t key_end = seqstore.layout.escape_first[position + 1];
         if (key_end > key_begin) {
            k_clear_keys<<<(key_end - key_begin + 255) / 256, 256, 0, hip_stream>>>(seqstore.layout.d_escapes, key_begin, key_end, dst_dev);
         }
         if (seqstore.dev.n_missing_runs > 0) {
            k_runs_clear_plane<<<(seqstore.dev.n_missing_runs + 255) / 256, 256, 0, hip_stream>>>(
               seqstore.dev.missing_run_keys, seqstore.dev.missing_run_ends, seqstore.dev.n_missing_runs, position, dst_dev
            );
         }
         const auto lo = std::lower_bound(seqstore.sparse_sorted.begin(), seqstore.sparse_sorted.end(), static_cast<uint64_t>(position) << 37);
         const auto hi = std::lower_bound(lo, seqstore.sparse_sorted.end(), (static_cast<uint64_t>(position) + 1) << 37);
         if (hi > lo) {
            const uint32_t begin = static_cast<uint32_t>(lo - seqstore.sparse_sorted.begin());
            const uint32_t end = static_cast<uint32_t>(hi - seqstore.sparse_sorted.begin());
            k_clear_keys<<<(end - begin + 255) / 256, 256, 0, hip_stream>>>(seqstore.d_sparse, begin, end, dst_dev);
         }
         HIP_TRY(hipGetLastError());
         return SILO_GPU_OK;
      }
      if (!seqstore.layout.escape_first_symbol.empty()) {  // rows of the symbol that are listed as escape keys (it has no code here)
         const size_t counter = static_cast<size_t>(position) * seqstore.dev.n_scan + seqstore.dev.index[symbol];
         const uint32_t begin = seqstore.layout.escape_first_symbol[counter];
         const uint32_t end = seqstore.layout.escape_first_symbol[counter + 1];
         if (end > begin) {
            k_scatter_sparse<<<(end - begin + 255) / 256, 256, 0, hip_stream>>>(seqstore.layout.d_escapes, begin, end, dst_dev);
            HIP_TRY(hipGetLastError());
         }
      }
      return SILO_GPU_OK;
   }
   if (seqstore.dev.kind[symbol] == PLANE_RUNS) {  // the missing symbol: its runs that cover the position
      HIP_TRY(hipMemsetAsync(dst_dev, 0, static_cast<size_t>(store->row_words) * sizeof(uint64_t), hip_stream));
      const uint32_t n_runs = seqstore.dev.n_missing_runs;
      if (n_runs > 0) {
         k_runs_to_plane<<<(n_runs + 255) / 256, 256, 0, hip_stream>>>(seqstore.dev.missing_run_keys, seqstore.dev.missing_run_ends, n_runs, position, dst_dev);
         HIP_TRY(hipGetLastError());
      }
      return SILO_GPU_OK;
   }
   if (seqstore.dev.kind[symbol] == PLANE_EXTRA) {  // already a plane: copy it
      HIP_TRY(hipMemcpyAsync(
         dst_dev, planePtr(seqstore.dev, position, symbol), static_cast<size_t>(store->row_words) * sizeof(uint64_t), hipMemcpyDeviceToDevice,
         hip_stream
      ));
      return SILO_GPU_OK;
   }
   HIP_TRY(hipMemsetAsync(dst_dev, 0, static_cast<size_t>(store->row_words) * sizeof(uint64_t), hip_stream));
   const uint64_t key_begin = (static_cast<uint64_t>(position) << 37) | (static_cast<uint64_t>(symbol) << 32);
   const uint64_t key_end = key_begin + (1ull << 32);
   const auto lo = std::lower_bound(seqstore.sparse_sorted.begin(), seqstore.sparse_sorted.end(), key_begin);
   const auto hi = std::lower_bound(lo, seqstore.sparse_sorted.end(), key_end);
   const uint32_t begin = static_cast<uint32_t>(lo - seqstore.sparse_sorted.begin());
   const uint32_t end = static_cast<uint32_t>(hi - seqstore.sparse_sorted.begin());
   if (end > begin) {
      k_scatter_sparse<<<(end - begin + 255) / 256, 256, 0, hip_stream>>>(seqstore.d_sparse, begin, end, dst_dev);
      HIP_TRY(hipGetLastError());
   }
   return SILO_GPU_OK;
}

struct silo_gpu_count_slot {
   unsigned long long* d_shards = nullptr;  // SILO_GPU_COUNT_SHARDS words + the tickets behind them
   uint32_t* d_ticket = nullptr;
   unsigned long long* host_total = nullptr;    // page-locked, written by the kernel
   unsigned long long* host_total_dev = nullptr;  // its device address
};

namespace {
constexpr unsigned long long COUNT_PENDING = ~0ull;
int filterEvalLaunch(
   const silo_gpu_store* store, const silo_gpu_bitprog* program, uint64_t* out_bitset_dev, uint64_t* out_count_dev, uint32_t* ticket_dev,
   unsigned long long* host_total_dev, void* stream
);
}  // namespace

int silo_gpu_filter_eval(const silo_gpu_store* store, const silo_gpu_bitprog* program, uint64_t* out_bitset_dev, uint64_t* out_count_dev, void* stream) {
   return filterEvalLaunch(store, program, out_bitset_dev, out_count_dev, nullptr, nullptr, stream);
}

int silo_gpu_count_slot_create(silo_gpu_count_slot** out_slot) {
   if (out_slot == nullptr) {
      return fail(SILO_GPU_ERR_INVALID_ARGUMENT, "silo_gpu_count_slot_create: null out pointer");
   }
   auto* slot = new (std::nothrow) silo_gpu_count_slot;
   if (slot == nullptr) {
      return fail(SILO_GPU_ERR_OUT_OF_MEMORY, "out of host memory");
   }
   // 64 count shards (64-bit), then the main ticket and one ticket per shard class (32-bit)
   const size_t device_bytes = SILO_GPU_COUNT_SHARDS * sizeof(unsigned long long) + (1 + SILO_GPU_COUNT_SHARDS) * sizeof(uint32_t);
   hipError_t err = hipMalloc(&slot->d_shards, device_bytes);
   if (err == hipSuccess) {
      err = hipMemset(slot->d_shards, 0, device_bytes);
   }
   if (err == hipSuccess) {
      // hipMemset of device memory only ENQUEUES a fill on the null stream, and the slot's first launch comes on a
      // non-blocking stream, which does not wait for the null stream: on a busy device the fill could land in the middle of
      // that launch and wipe tickets already taken ("the kernel finished without delivering its total")
      err = hipStreamSynchronize(nullptr);
   }
   if (err == hipSuccess) {
      slot->d_ticket = reinterpret_cast<uint32_t*>(slot->d_shards + SILO_GPU_COUNT_SHARDS);
      err = hipHostMalloc(&slot->host_total, sizeof(unsigned long long), hipHostMallocMapped | hipHostMallocCoherent);
   }
   if (err == hipSuccess) {
      *slot->host_total = COUNT_PENDING;
      err = hipHostGetDevicePointer(reinterpret_cast<void**>(&slot->host_total_dev), slot->host_total, 0);
   }
   if (err != hipSuccess) {
      silo_gpu_count_slot_destroy(slot);
      HIP_TRY(err);
   }
   *out_slot = slot;
   return SILO_GPU_OK;
}

void silo_gpu_count_slot_destroy(silo_gpu_count_slot* slot) {
   if (slot != nullptr) {
      (void)hipFree(slot->d_shards);
      if (slot->host_total != nullptr) {
         (void)hipHostFree(slot->host_total);
      }
      delete slot;
   }
}

int silo_gpu_filter_eval_count(
   const silo_gpu_store* store, const silo_gpu_bitprog* program, uint64_t* out_bitset_dev, silo_gpu_count_slot* slot, void* stream
) {
   if (slot == nullptr) {
      return fail(SILO_GPU_ERR_INVALID_ARGUMENT, "silo_gpu_filter_eval_count: null slot");
   }
   __atomic_store_n(slot->host_total, COUNT_PENDING, __ATOMIC_RELEASE);
   return filterEvalLaunch(
      store, program, out_bitset_dev, reinterpret_cast<uint64_t*>(slot->d_shards), slot->d_ticket, slot->host_total_dev, stream
   );
}

int silo_gpu_count_slot_wait(silo_gpu_count_slot* slot, uint64_t* out_count, void* stream) {
   if (slot == nullptr || out_count == nullptr) {
      return fail(SILO_GPU_ERR_INVALID_ARGUMENT, "silo_gpu_count_slot_wait: null argument");
   }
   // The kernel's last block stores the total with system scope.  The wait is a pure spin on that word — no HIP call from
   // the polling threads (round 1 polled hipStreamQuery from every request thread; under rocprofv3's kernel tracing that run
   // segfaulted, and whether the fault was the profiler's or the polling's was never established, so the polling is gone).
   // A launch that does not deliver within the spin budget (tens of milliseconds: a failed or wedged launch, or a very busy
   // device) is waited for with ONE blocking hipStreamSynchronize, which also reports a broken stream.
   constexpr uint64_t SPIN_BUDGET = uint64_t{1} << 20;
   for (uint64_t spin = 0; spin < SPIN_BUDGET; ++spin) {
      const unsigned long long value = __atomic_load_n(slot->host_total, __ATOMIC_ACQUIRE);
      if (value != COUNT_PENDING) {
         *out_count = value;
         return SILO_GPU_OK;
      }
#if defined(__x86_64__)
      __builtin_ia32_pause();
#endif
   }
   HIP_TRY(hipStreamSynchronize(static_cast<hipStream_t>(stream)));
   const unsigned long long value = __atomic_load_n(slot->host_total, __ATOMIC_ACQUIRE);
   if (value == COUNT_PENDING) {
      return fail(SILO_GPU_ERR_HIP, "count slot: the kernel finished without delivering its total");
   }
   *out_count = value;
   return SILO_GPU_OK;
}

namespace {
/// Limits and operands of a bit-program, checked on the host: a bad slot or leaf index would be an out-of-bounds LDS /
/// global access on the device.
int validateProgram(const silo_gpu_bitprog* program) {
   if (program == nullptr || program->code == nullptr) {
      return fail(SILO_GPU_ERR_INVALID_ARGUMENT, "silo_gpu_filter_eval: null program");
   }
   if (program->n_instructions == 0 || program->n_instructions > SILO_GPU_MAX_INSTRUCTIONS ||
       program->n_leaves > SILO_GPU_MAX_LEAVES || program->n_slots == 0 || program->n_slots > SILO_GPU_MAX_SLOTS) {
      return fail(SILO_GPU_ERR_PROGRAM_TOO_LARGE, "bit-program exceeds the instruction / leaf / slot limits");
   }
   // validate operands on the host: a bad slot or leaf index would be an out-of-bounds LDS / global access
   for (uint32_t pc = 0; pc < program->n_instructions; ++pc) {
      const uint32_t w0 = program->code[2 * pc];
      const uint32_t imm = program->code[2 * pc + 1];
      const uint32_t op = w0 & 0xFFu, dst = (w0 >> 8) & 0xFFu, a = (w0 >> 16) & 0xFFu, b = (w0 >> 24) & 0xFFu;
      bool ok = true;
      // a readable operand is a slot or, from SILO_GPU_LEAF_OPERAND up, a leaf
      const auto readable = [&](uint32_t operand) {
         return operand < program->n_slots ||
                (operand >= SILO_GPU_LEAF_OPERAND && operand - SILO_GPU_LEAF_OPERAND < program->n_leaves);
      };
      switch (op) {
         case SILO_GPU_OP_LOAD:
            ok = dst < program->n_slots && imm < program->n_leaves;
            break;
         case SILO_GPU_OP_ZERO:
         case SILO_GPU_OP_ONES:
            ok = dst < program->n_slots;
            break;
         case SILO_GPU_OP_NOT:
         case SILO_GPU_OP_MOV:
            ok = dst < program->n_slots && readable(a);
            break;
         case SILO_GPU_OP_AND:
         case SILO_GPU_OP_OR:
         case SILO_GPU_OP_ANDNOT:
            ok = dst < program->n_slots && readable(a) && readable(b);
            break;
         case SILO_GPU_OP_CNT_ADD:
            ok = readable(a) && b >= 1 && dst + b <= program->n_slots;
            break;
         case SILO_GPU_OP_OR_N:
         case SILO_GPU_OP_AND_N:
            ok = dst < program->n_slots && (imm >> 16) >= 1 && (imm & 0xFFFFu) + (imm >> 16) <= program->n_leaves;
            break;
         case SILO_GPU_OP_CNT_ADD_N:
         case SILO_GPU_OP_CNT_ADD_NOT_N:
            ok = b >= 1 && dst + b <= program->n_slots && (imm >> 16) >= 1 && (imm & 0xFFFFu) + (imm >> 16) <= program->n_leaves;
            break;
         case SILO_GPU_OP_CNT_GE:
         case SILO_GPU_OP_CNT_EQ:
            ok = dst < program->n_slots && b >= 1 && a + b <= program->n_slots;
            break;
         default:
            ok = false;
      }
      if (!ok) {
         return fail(SILO_GPU_ERR_INVALID_ARGUMENT, "bit-program instruction " + std::to_string(pc) + " has an invalid operand");
      }
   }
   for (uint32_t k = 0; k < program->n_leaves; ++k) {
      if (program->leaves == nullptr || program->leaves[k] == nullptr) {
         return fail(SILO_GPU_ERR_INVALID_ARGUMENT, "bit-program leaf " + std::to_string(k) + " is null");
      }
   }
   return SILO_GPU_OK;
}

int filterEvalLaunch(
   const silo_gpu_store* store, const silo_gpu_bitprog* program, uint64_t* out_bitset_dev, uint64_t* out_count_dev, uint32_t* ticket_dev,
   unsigned long long* host_total_dev, void* stream
) {
   if (store == nullptr || program == nullptr || program->code == nullptr) {
      return fail(SILO_GPU_ERR_INVALID_ARGUMENT, "silo_gpu_filter_eval: bad arguments");
   }
   HIP_TRY(hipSetDevice(store->device));  // a new host thread starts on device 0
   if (const int rc = validateProgram(program); rc != SILO_GPU_OK) {
      return rc;
   }
   FilterEvalArgs args{};
   args.n_instructions = program->n_instructions;
   args.sequence_count = store->sequence_count;
   args.row_words = store->row_words;
   args.n_slots = program->n_slots;
   args.out = out_bitset_dev;
   args.out_count = reinterpret_cast<unsigned long long*>(out_count_dev);
   args.ticket = ticket_dev;
   args.host_total = host_total_dev;
   for (uint32_t k = 0; k < program->n_leaves; ++k) {
      args.leaves[k] = program->leaves[k];
   }
   memcpy(args.code, program->code, static_cast<size_t>(program->n_instructions) * 2 * sizeof(uint32_t));
   const uint32_t blocks = (store->row_words + EVAL_WORDS_PER_BLOCK - 1) / EVAL_WORDS_PER_BLOCK;
   const size_t lds_bytes = static_cast<size_t>(program->n_slots) * EVAL_THREADS * sizeof(ulonglong2);  // <= 32 KiB
   if (g_tune_eval_leaf_batch.load() == 16) {
      k_filter_eval<16><<<blocks, EVAL_THREADS, lds_bytes, static_cast<hipStream_t>(stream)>>>(args);
   } else {
      k_filter_eval<8><<<blocks, EVAL_THREADS, lds_bytes, static_cast<hipStream_t>(stream)>>>(args);
   }
   HIP_TRY(hipGetLastError());
   return SILO_GPU_OK;
}
}  // namespace

namespace {
/// Staging of silo_gpu_filter_eval_batch: the program table in page-locked host memory and on the device, the count
/// shards on the device and their page-locked landing area.  One per host thread; every call ends with a stream
/// synchronisation, so a buffer is never reused while the device still reads it.  Never freed (thread exit may come
/// after the HIP runtime has shut down).
struct EvalBatchScratch {
   uint8_t* host_table = nullptr;
   uint8_t* device_table = nullptr;
   size_t table_capacity = 0;
   uint32_t* host_counts = nullptr;
   size_t counts_capacity = 0;  // programs
};
}  // namespace

int silo_gpu_filter_eval_batch(
   const silo_gpu_store* store, const silo_gpu_bitprog* programs, uint32_t n_programs, uint64_t* const* out_bitsets_dev, uint64_t* out_counts,
   void* stream
) {
   if (store == nullptr || (n_programs != 0 && programs == nullptr)) {
      return fail(SILO_GPU_ERR_INVALID_ARGUMENT, "silo_gpu_filter_eval_batch: bad arguments");
   }
   if (n_programs == 0) {
      return SILO_GPU_OK;
   }
   HIP_TRY(hipSetDevice(store->device));
   for (uint32_t q = 0; q < n_programs; ++q) {
      if (const int rc = validateProgram(&programs[q]); rc != SILO_GPU_OK) {
         return rc;
      }
   }
   // programs with few slots first: a launch sizes its LDS for the hungriest program in it, so the (typical) programs
   // with a handful of slots are not held to the occupancy of a rare wide one
   std::vector<uint32_t> order(n_programs);
   for (uint32_t q = 0; q < n_programs; ++q) {
      order[q] = q;
   }
   std::stable_sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return programs[a].n_slots < programs[b].n_slots; });
   const auto align16 = [](size_t value) { return (value + 15) / 16 * 16; };
   size_t table_bytes = align16(static_cast<size_t>(n_programs) * sizeof(BatchProgramHeader));
   for (uint32_t q = 0; q < n_programs; ++q) {
      table_bytes += align16(static_cast<size_t>(programs[q].n_instructions) * 2 * sizeof(uint32_t)) + align16(static_cast<size_t>(programs[q].n_leaves) * sizeof(uint64_t*));
   }
   // the count shards travel at the end of the table: the upload that brings the programs also zeroes them (no memset
   // launch), and their device copy is read back from the same allocation
   const size_t counts_bytes = static_cast<size_t>(n_programs) * EVAL_BATCH_SHARDS * sizeof(uint32_t);
   const size_t counts_offset = table_bytes;
   table_bytes += align16(counts_bytes);
   thread_local EvalBatchScratch scratch;
   if (table_bytes > scratch.table_capacity) {
      if (scratch.host_table != nullptr) {
         (void)hipHostFree(scratch.host_table);
         (void)hipFree(scratch.device_table);
         scratch.host_table = nullptr;
         scratch.device_table = nullptr;
         scratch.table_capacity = 0;
      }
      const size_t capacity = std::max<size_t>(table_bytes * 2, size_t{64} << 10);
      HIP_TRY(hipHostMalloc(&scratch.host_table, capacity, hipHostMallocDefault));
      HIP_TRY(hipMalloc(&scratch.device_table, capacity));
      scratch.table_capacity = capacity;
   }
   if (n_programs > scratch.counts_capacity) {
      if (scratch.host_counts != nullptr) {
         (void)hipHostFree(scratch.host_counts);
         scratch.host_counts = nullptr;
         scratch.counts_capacity = 0;
      }
      const size_t capacity = std::max<size_t>(static_cast<size_t>(n_programs) * 2, 128);
      HIP_TRY(hipHostMalloc(&scratch.host_counts, capacity * EVAL_BATCH_SHARDS * sizeof(uint32_t), hipHostMallocDefault));
      scratch.counts_capacity = capacity;
   }
   memset(scratch.host_table + counts_offset, 0, counts_bytes);
   auto* headers = reinterpret_cast<BatchProgramHeader*>(scratch.host_table);
   size_t cursor = align16(static_cast<size_t>(n_programs) * sizeof(BatchProgramHeader));
   for (uint32_t slot = 0; slot < n_programs; ++slot) {  // table slot `slot` holds program order[slot]
      const silo_gpu_bitprog& program = programs[order[slot]];
      BatchProgramHeader& header = headers[slot];
      header.n_instructions = program.n_instructions;
      header.n_slots = program.n_slots;
      header.code_offset = static_cast<uint32_t>(cursor);
      memcpy(scratch.host_table + cursor, program.code, static_cast<size_t>(program.n_instructions) * 2 * sizeof(uint32_t));
      cursor += align16(static_cast<size_t>(program.n_instructions) * 2 * sizeof(uint32_t));
      header.leaf_offset = static_cast<uint32_t>(cursor);
      if (program.n_leaves != 0) {
         memcpy(scratch.host_table + cursor, program.leaves, static_cast<size_t>(program.n_leaves) * sizeof(uint64_t*));
      }
      cursor += align16(static_cast<size_t>(program.n_leaves) * sizeof(uint64_t*));
      header.out = out_bitsets_dev != nullptr ? out_bitsets_dev[order[slot]] : nullptr;
      header.reserved = 0;
   }
   auto hip_stream = static_cast<hipStream_t>(stream);
   HIP_TRY(hipMemcpyAsync(scratch.device_table, scratch.host_table, table_bytes, hipMemcpyHostToDevice, hip_stream));
   uint32_t* device_counts = reinterpret_cast<uint32_t*>(scratch.device_table + counts_offset);
   static std::once_flag lds_once;
   std::call_once(lds_once, [] {
      // 4 waves x up to 32 slots x 1 KiB: beyond the 64 KiB a kernel may ask for by default
      (void)hipFuncSetAttribute(
         reinterpret_cast<const void*>(k_filter_eval_batch), hipFuncAttributeMaxDynamicSharedMemorySize,
         (EVAL_BATCH_THREADS / 64) * SILO_GPU_MAX_SLOTS * 64 * static_cast<int>(sizeof(ulonglong2))
      );
   });
   const uint32_t tiles = (store->row_words + EVAL_BATCH_THREADS * 2 - 1) / (EVAL_BATCH_THREADS * 2);
   for (uint32_t first = 0; first < n_programs;) {  // one launch per slot class: <= 8, <= 16, <= 32 slots
      const uint32_t class_slots = programs[order[first]].n_slots <= 8 ? 8 : (programs[order[first]].n_slots <= 16 ? 16 : SILO_GPU_MAX_SLOTS);
      uint32_t last = first;
      uint32_t max_slots = 1;
      while (last < n_programs && programs[order[last]].n_slots <= class_slots) {
         max_slots = std::max(max_slots, programs[order[last]].n_slots);
         ++last;
      }
      const size_t lds_bytes = static_cast<size_t>(EVAL_BATCH_THREADS / 64) * max_slots * 64 * sizeof(ulonglong2);
      k_filter_eval_batch<<<dim3(last - first, tiles), EVAL_BATCH_THREADS, lds_bytes, hip_stream>>>(
         scratch.device_table, first, store->sequence_count, store->row_words, max_slots, device_counts
      );
      HIP_TRY(hipGetLastError());
      first = last;
   }
   HIP_TRY(hipMemcpyAsync(scratch.host_counts, device_counts, counts_bytes, hipMemcpyDeviceToHost, hip_stream));
   HIP_TRY(hipStreamSynchronize(hip_stream));
   if (out_counts != nullptr) {
      for (uint32_t slot = 0; slot < n_programs; ++slot) {
         uint64_t total = 0;
         for (uint32_t shard = 0; shard < EVAL_BATCH_SHARDS; ++shard) {
            total += scratch.host_counts[static_cast<size_t>(slot) * EVAL_BATCH_SHARDS + shard];
         }
         out_counts[order[slot]] = total;
      }
   }
   return SILO_GPU_OK;
}

int silo_gpu_popcount(const silo_gpu_store* store, const uint64_t* bitset_dev, uint64_t* out_count_dev, void* stream) {
   if (store == nullptr || bitset_dev == nullptr || out_count_dev == nullptr) {
      return fail(SILO_GPU_ERR_INVALID_ARGUMENT, "silo_gpu_popcount: bad arguments");
   }
   HIP_TRY(hipSetDevice(store->device));  // a new host thread starts on device 0
   const uint32_t chunks = store->row_words / 2;
   const uint32_t blocks = std::min<uint32_t>((chunks + 255) / 256, 1024u);
   k_popcount<<<blocks, 256, 0, static_cast<hipStream_t>(stream)>>>(
      bitset_dev, store->row_words, reinterpret_cast<unsigned long long*>(out_count_dev)
   );
   HIP_TRY(hipGetLastError());
   return SILO_GPU_OK;
}

int silo_gpu_mutations_scan(
   const silo_gpu_store* store, uint32_t seqstore_id, const uint64_t* filter_dev, uint32_t pos_begin, uint32_t pos_end,
   uint32_t* counts_out_dev, void* stream
) {
   if (store == nullptr || seqstore_id >= store->seqstores.size() || counts_out_dev == nullptr) {
      return fail(SILO_GPU_ERR_INVALID_ARGUMENT, "silo_gpu_mutations_scan: bad arguments");
   }
   HIP_TRY(hipSetDevice(store->device));
   const SeqStoreDev& dev = store->seqstores[seqstore_id].dev;
   if (pos_begin > pos_end || pos_end > dev.positions) {
      return fail(SILO_GPU_ERR_INVALID_ARGUMENT, "position range out of bounds");
   }
   if (pos_begin == pos_end || dev.n_scan == 0) {
      return SILO_GPU_OK;
   }
   auto hip_stream_early = static_cast<hipStream_t>(stream);
   if (filter_dev == nullptr) {
      // Full filter: add the cached totals of the unfiltered store instead of streaming the planes again.
      auto* mutable_store = const_cast<silo_gpu_store*>(store);  // the cache is logically const
      SeqStoreHost& seqstore = mutable_store->seqstores[seqstore_id];
      const size_t n_totals = static_cast<size_t>(dev.positions) * dev.n_scan;
      {
         const std::lock_guard<std::mutex> lock(mutable_store->mutex);
         if (!seqstore.totals_ready) {
            if (seqstore.d_totals == nullptr) {
               HIP_TRY(hipMalloc(&seqstore.d_totals, n_totals * sizeof(uint32_t)));
            }
            HIP_TRY(hipMemsetAsync(seqstore.d_totals, 0, n_totals * sizeof(uint32_t), hip_stream_early));
            const int rc = silo_gpu_mutations_scan(store, seqstore_id, store->d_ones, 0, dev.positions, seqstore.d_totals, stream);
            if (rc != SILO_GPU_OK) {
               return rc;
            }
            HIP_TRY(hipStreamSynchronize(hip_stream_early));  // other streams may read it from now on
            seqstore.totals_ready = true;
         }
      }
      const uint32_t n = (pos_end - pos_begin) * dev.n_scan;
      k_add_u32<<<(n + 255) / 256, 256, 0, hip_stream_early>>>(
         counts_out_dev, seqstore.d_totals + static_cast<size_t>(pos_begin) * dev.n_scan, n
      );
      HIP_TRY(hipGetLastError());
      g_last_scan_kernel = "k_add_u32 (cached totals)";
      return SILO_GPU_OK;
   }
   const silo_gpu_scan_range range{seqstore_id, pos_begin, pos_end};
   return silo_gpu_mutations_scan_ranges(store, &range, 1, &filter_dev, 1, &counts_out_dev, stream);
}

int silo_gpu_reconstruct_sequences(
   const silo_gpu_store* store, uint32_t seqstore_id, const uint32_t* row_ids_dev, uint32_t n_rows, char* out_chars_dev, void* stream
) {
   if (store == nullptr || seqstore_id >= store->seqstores.size() || row_ids_dev == nullptr || out_chars_dev == nullptr) {
      return fail(SILO_GPU_ERR_INVALID_ARGUMENT, "silo_gpu_reconstruct_sequences: bad arguments");
   }
   if (n_rows > 65535) {
      return fail(SILO_GPU_ERR_INVALID_ARGUMENT, "silo_gpu_reconstruct_sequences: at most 65535 rows per call");
   }
   HIP_TRY(hipSetDevice(store->device));
   auto* mutable_store = const_cast<silo_gpu_store*>(store);  // the symbol -> char table is created on first use
   const SeqStoreHost& seqstore = store->seqstores[seqstore_id];
   if (!seqstore.finalized) {
      return fail(SILO_GPU_ERR_INVALID_ARGUMENT, "silo_gpu_reconstruct_sequences: store is not finalized");
   }
   if (n_rows == 0 || seqstore.dev.positions == 0) {
      return SILO_GPU_OK;
   }
   const uint32_t alphabet = seqstore.alphabet == SILO_GPU_ALPHABET_AMINO_ACID ? 1 : 0;
   {
      const std::lock_guard<std::mutex> lock(mutable_store->mutex);
      if (mutable_store->d_symbol_chars[alphabet] == nullptr) {
         // enum order of the reference's alphabets (nucleotide_symbols.h:15-34, aa_symbols.h:15-43)
         const char* chars = alphabet == 0 ? "-ACGTRYSWKMBDHVN" : "-ACDEFGHIKLMNPQRSTVWYBZ*X";
         char* device = nullptr;
         HIP_TRY(hipMalloc(&device, SILO_GPU_MAX_SYMBOLS));
         HIP_TRY(hipMemcpy(device, chars, strlen(chars), hipMemcpyHostToDevice));
         mutable_store->d_symbol_chars[alphabet] = device;
      }
   }
   const dim3 grid((seqstore.dev.positions + 255) / 256, n_rows);
   k_reconstruct_sequences<<<grid, 256, 0, static_cast<hipStream_t>(stream)>>>(
      seqstore.dev, seqstore.d_sparse, static_cast<uint32_t>(seqstore.sparse_sorted.size()), row_ids_dev,
      store->d_symbol_chars[alphabet], out_chars_dev
   );
   HIP_TRY(hipGetLastError());
   return SILO_GPU_OK;
}

int silo_gpu_mutations_select(
   const uint32_t* counts_dev, const uint8_t* reference_index_dev, uint32_t n_positions, uint32_t n_symbols, double min_proportion,
   uint32_t capacity, uint32_t* out_dev, void* stream
) {
   if (counts_dev == nullptr || reference_index_dev == nullptr || out_dev == nullptr || n_symbols == 0 || n_symbols > 32) {
      return fail(SILO_GPU_ERR_INVALID_ARGUMENT, "silo_gpu_mutations_select: bad arguments");
   }
   auto hip_stream = static_cast<hipStream_t>(stream);
   HIP_TRY(hipMemsetAsync(out_dev, 0, 4 * sizeof(uint32_t), hip_stream));
   if (n_positions != 0) {
      k_mutations_select<<<(n_positions + 255) / 256, 256, 0, hip_stream>>>(
         counts_dev, reference_index_dev, n_positions, n_symbols, min_proportion, capacity, out_dev
      );
      HIP_TRY(hipGetLastError());
   }
   return SILO_GPU_OK;
}

struct silo_gpu_row_slot {
   uint32_t capacity = 0;
   uint32_t epoch = 0;                       // of the last launch
   uint32_t* d_cursor_and_ticket = nullptr;  // device: rows appended so far, blocks done so far
   void* host = nullptr;                     // page-locked: header word (epoch << 32 | selected cells), then the rows from byte 16
   void* host_dev = nullptr;                 // its device address
};

int silo_gpu_row_slot_create(uint32_t row_capacity, silo_gpu_row_slot** out_slot) {
   if (out_slot == nullptr || row_capacity == 0) {
      return fail(SILO_GPU_ERR_INVALID_ARGUMENT, "silo_gpu_row_slot_create: bad arguments");
   }
   auto* slot = new (std::nothrow) silo_gpu_row_slot;
   if (slot == nullptr) {
      return fail(SILO_GPU_ERR_OUT_OF_MEMORY, "out of host memory");
   }
   slot->capacity = row_capacity;
   hipError_t err = hipMalloc(&slot->d_cursor_and_ticket, 2 * sizeof(uint32_t));
   err = err != hipSuccess ? err : hipMemset(slot->d_cursor_and_ticket, 0, 2 * sizeof(uint32_t));
   err = err != hipSuccess ? err : hipStreamSynchronize(nullptr);  // (the fill is only enqueued; the launches come on other streams)
   err = err != hipSuccess ? err : hipHostMalloc(&slot->host, 16 + sizeof(silo_gpu_mutation_row) * static_cast<size_t>(row_capacity), hipHostMallocMapped | hipHostMallocCoherent);
   if (err == hipSuccess) {
      *static_cast<unsigned long long*>(slot->host) = 0;
      err = hipHostGetDevicePointer(&slot->host_dev, slot->host, 0);
   }
   if (err != hipSuccess) {
      silo_gpu_row_slot_destroy(slot);
      HIP_TRY(err);
   }
   *out_slot = slot;
   return SILO_GPU_OK;
}

void silo_gpu_row_slot_destroy(silo_gpu_row_slot* slot) {
   if (slot != nullptr) {
      (void)hipFree(slot->d_cursor_and_ticket);
      if (slot->host != nullptr) {
         (void)hipHostFree(slot->host);
      }
      delete slot;
   }
}

int silo_gpu_mutations_select_to_slot(
   const uint32_t* counts_dev, const uint8_t* reference_index_dev, uint32_t n_positions, uint32_t n_symbols, double min_proportion,
   silo_gpu_row_slot* slot, void* stream
) {
   if (counts_dev == nullptr || reference_index_dev == nullptr || slot == nullptr || n_symbols == 0 || n_symbols > 32 || n_positions == 0) {
      return fail(SILO_GPU_ERR_INVALID_ARGUMENT, "silo_gpu_mutations_select_to_slot: bad arguments");
   }
   slot->epoch += 1;
   if (slot->epoch == 0) {
      slot->epoch = 1;
   }
   auto* header = static_cast<unsigned long long*>(slot->host_dev);
   k_mutations_select_to_host<<<(n_positions + 255) / 256, 256, 0, static_cast<hipStream_t>(stream)>>>(
      counts_dev, reference_index_dev, n_positions, n_symbols, min_proportion, slot->capacity, slot->d_cursor_and_ticket,
      reinterpret_cast<silo_gpu_mutation_row*>(reinterpret_cast<char*>(slot->host_dev) + 16), header, slot->epoch
   );
   HIP_TRY(hipGetLastError());
   return SILO_GPU_OK;
}

int silo_gpu_row_slot_wait(silo_gpu_row_slot* slot, const silo_gpu_mutation_row** out_rows, uint32_t* out_selected, void* stream) {
   if (slot == nullptr || out_rows == nullptr || out_selected == nullptr) {
      return fail(SILO_GPU_ERR_INVALID_ARGUMENT, "silo_gpu_row_slot_wait: null argument");
   }
   const auto* header = static_cast<const unsigned long long*>(slot->host);
   // a pure spin on the header word (as silo_gpu_count_slot_wait); a launch that does not deliver within the budget is waited
   // for with ONE blocking hipStreamSynchronize, which also reports a broken stream
   constexpr uint64_t SPIN_BUDGET = uint64_t{1} << 22;
   unsigned long long value = 0;
   bool delivered = false;
   for (uint64_t spin = 0; spin < SPIN_BUDGET && !delivered; ++spin) {
      value = __atomic_load_n(header, __ATOMIC_ACQUIRE);
      delivered = static_cast<uint32_t>(value >> 32) == slot->epoch;
#if defined(__x86_64__)
      if (!delivered) {
         __builtin_ia32_pause();
      }
#endif
   }
   if (!delivered) {
      HIP_TRY(hipStreamSynchronize(static_cast<hipStream_t>(stream)));
      value = __atomic_load_n(header, __ATOMIC_ACQUIRE);
      if (static_cast<uint32_t>(value >> 32) != slot->epoch) {
         return fail(SILO_GPU_ERR_HIP, "row slot: the kernel finished without delivering its rows");
      }
   }
   *out_selected = static_cast<uint32_t>(value);
   *out_rows = reinterpret_cast<const silo_gpu_mutation_row*>(static_cast<const char*>(slot->host) + 16);
   return SILO_GPU_OK;
}

namespace {
/// Plain stream read: the sum of `n_chunks` 16-byte chunks, 16 non-temporal loads in flight per lane, every block a contiguous
/// stretch of the buffer — the achievable HBM read rate that SURVEY.md section 8(d) asks the scan to be compared with.
__global__ __launch_bounds__(256) void k_stream_sum(const uint64_t* __restrict__ data, uint64_t n_chunks, unsigned long long* __restrict__ sink) {
   constexpr uint32_t IN_FLIGHT = 16;
   const uint64_t per_block = (n_chunks + gridDim.x - 1) / gridDim.x;
   const uint64_t begin = static_cast<uint64_t>(blockIdx.x) * per_block;
   const uint64_t end = min(n_chunks, begin + per_block);
   uint64_t sum = 0;
   for (uint64_t base = begin; base < end; base += 256u * IN_FLIGHT) {
      ulonglong2 value[IN_FLIGHT];
#pragma unroll
      for (uint32_t k = 0; k < IN_FLIGHT; ++k) {
         const uint64_t chunk = base + k * 256u + threadIdx.x;
         value[k] = chunk < end ? loadPlane16<true>(data + chunk * 2u) : make_ulonglong2(0, 0);
      }
#pragma unroll
      for (uint32_t k = 0; k < IN_FLIGHT; ++k) {
         sum += value[k].x + value[k].y;
      }
   }
   if (sum == 0x123456789ABCDEFull) {  // (never: keeps the loads alive without a store per thread)
      atomicAdd(sink, 1ull);
   }
}
}  // namespace

int silo_gpu_stream_read_probe(uint64_t bytes, uint32_t reps, float* out_ms_per_pass) {
   if (out_ms_per_pass == nullptr || bytes < (1u << 20) || reps == 0) {
      return fail(SILO_GPU_ERR_INVALID_ARGUMENT, "silo_gpu_stream_read_probe: bad arguments");
   }
   uint64_t* data = nullptr;
   unsigned long long* sink = nullptr;
   hipEvent_t start = nullptr, stop = nullptr;
   bytes &= ~uint64_t{15};
   hipError_t status = hipMalloc(&data, bytes);
   status = status != hipSuccess ? status : hipMalloc(&sink, sizeof(unsigned long long));
   status = status != hipSuccess ? status : hipMemset(data, 0x5A, bytes);
   status = status != hipSuccess ? status : hipMemset(sink, 0, sizeof(unsigned long long));
   status = status != hipSuccess ? status : hipEventCreate(&start);
   status = status != hipSuccess ? status : hipEventCreate(&stop);
   float ms = 0;
   if (status == hipSuccess) {
      const uint32_t blocks = 256u * 32u;  // many short blocks: a few rounds of 4-8 blocks of 256 threads per CU
      k_stream_sum<<<blocks, 256>>>(data, bytes / 16, sink);  // warm-up
      status = hipEventRecord(start, nullptr);
      for (uint32_t rep = 0; rep < reps && status == hipSuccess; ++rep) {
         k_stream_sum<<<blocks, 256>>>(data, bytes / 16, sink);
         status = hipGetLastError();
      }
      status = status != hipSuccess ? status : hipEventRecord(stop, nullptr);
      status = status != hipSuccess ? status : hipEventSynchronize(stop);
      status = status != hipSuccess ? status : hipEventElapsedTime(&ms, start, stop);
   }
   (void)hipFree(data);
   (void)hipFree(sink);
   if (start != nullptr) {
      (void)hipEventDestroy(start);
   }
   if (stop != nullptr) {
      (void)hipEventDestroy(stop);
   }
   HIP_TRY(status);
   *out_ms_per_pass = ms / static_cast<float>(reps);
   return SILO_GPU_OK;
}

int silo_gpu_upload_bytes(const void* src_host, size_t bytes, void** out_dev) {
   if (src_host == nullptr || out_dev == nullptr || bytes == 0) {
      return fail(SILO_GPU_ERR_INVALID_ARGUMENT, "silo_gpu_upload_bytes: bad arguments");
   }
   void* dev = nullptr;
   HIP_TRY(hipMalloc(&dev, bytes));
   const hipError_t status = hipMemcpy(dev, src_host, bytes, hipMemcpyHostToDevice);
   if (status != hipSuccess) {
      (void)hipFree(dev);
      HIP_TRY(status);
   }
   *out_dev = dev;
   return SILO_GPU_OK;
}

}  // extern "C"

// ================================================================================================
// Import from the reference's own storage form (the payload half): a Position is a roaring bitmap per
// symbol (position.h:27-37) — in a snapshot each written in CRoaring's PORTABLE serialization (roaring_serialize.h:17-45) —
// where the most numerous symbol is stored flipped (its complement) or deleted (empty) (position.cpp:42-127), and the missing
// symbol lives ROW-wise in missing_symbol_bitmaps (sequence_store.cpp:153-190).  The payloads are expanded on the device:
// the host only reads the directory of a bitmap (cookie, container keys / cardinalities / offsets); one block per roaring
// container then turns an array, bitset or run container into its 2^16-row slice of a dense row, which is merged into the
// build-time planes exactly where append_sequences would have put those rows.  The Boost archive framing AROUND the payloads
// (database.cpp:369-461) is not read: no Boost here to pin it against — DESIGN.md §10.
//
// Portable format (published CRoaring specification; restated, the library itself is not in the image — parity unpinned):
//   uint32 cookie: low 16 bits 12347 -> run containers may occur, n = (cookie >> 16) + 1, then ceil(n / 8) bytes of run flags;
//                  12346 -> no run containers, then uint32 n;
//   n x (uint16 key, uint16 cardinality - 1);
//   n x uint32 offset of the container's data, present unless (cookie 12347 and n < 4);
//   per container: array (cardinality <= 4096, not run): cardinality x uint16 ascending; bitset: 1024 x uint64;
//                  run: uint16 n_runs, n_runs x (uint16 start, uint16 length - 1).
// ================================================================================================
namespace {

enum : uint32_t { ROARING_ARRAY = 0, ROARING_BITSET = 1, ROARING_RUN = 2 };

struct RoaringContainer {
   uint32_t key;
   uint32_t type;
   uint32_t count;   // array: values, run: runs
   uint32_t offset;  // bytes into the payload (any alignment)
};

int parseRoaringDirectory(const uint8_t* bytes, size_t n_bytes, std::vector<RoaringContainer>& out) {
   out.clear();
   const auto bad = [](const char* what) { return fail(SILO_GPU_ERR_INVALID_ARGUMENT, std::string("roaring payload: ") + what); };
   const auto u16 = [&](size_t at) { return static_cast<uint32_t>(bytes[at]) | (static_cast<uint32_t>(bytes[at + 1]) << 8); };
   const auto u32 = [&](size_t at) { return u16(at) | (u16(at + 2) << 16); };
   if (n_bytes < 8) {
      return n_bytes == 0 ? SILO_GPU_OK : bad("shorter than its header");
   }
   const uint32_t cookie = u32(0);
   size_t cursor = 4;
   uint32_t n = 0;
   const uint8_t* run_flags = nullptr;
   if ((cookie & 0xFFFFu) == 12347u) {
      n = (cookie >> 16) + 1;
      run_flags = bytes + cursor;
      cursor += (n + 7) / 8;
   } else if (cookie == 12346u) {
      n = u32(cursor);
      cursor += 4;
   } else {
      return bad("unknown cookie");
   }
   if (n > 65536 || cursor + static_cast<size_t>(n) * 4 > n_bytes) {
      return bad("container count does not fit the payload");
   }
   const size_t descriptors = cursor;
   cursor += static_cast<size_t>(n) * 4;
   const bool has_offsets = run_flags == nullptr || n >= 4;
   const size_t offsets = cursor;
   if (has_offsets) {
      cursor += static_cast<size_t>(n) * 4;
      if (cursor > n_bytes) {
         return bad("offset header does not fit the payload");
      }
   }
   out.reserve(n);
   for (uint32_t k = 0; k < n; ++k) {
      RoaringContainer container{};
      container.key = u16(descriptors + static_cast<size_t>(k) * 4);
      const uint32_t cardinality = u16(descriptors + static_cast<size_t>(k) * 4 + 2) + 1;
      const bool is_run = run_flags != nullptr && ((run_flags[k / 8] >> (k % 8)) & 1u) != 0;
      size_t data = has_offsets ? u32(offsets + static_cast<size_t>(k) * 4) : cursor;
      size_t size = 0;
      if (is_run) {
         if (data + 2 > n_bytes) {
            return bad("run container past the end");
         }
         container.type = ROARING_RUN;
         container.count = u16(data);
         data += 2;
         size = static_cast<size_t>(container.count) * 4;
      } else if (cardinality <= 4096) {
         container.type = ROARING_ARRAY;
         container.count = cardinality;
         size = static_cast<size_t>(cardinality) * 2;
      } else {
         container.type = ROARING_BITSET;
         container.count = 1024;
         size = 8192;
      }
      if (data + size > n_bytes) {  // (no alignment to expect: the run flags take ceil(n / 8) bytes)
         return bad("container data past the end");
      }
      container.offset = static_cast<uint32_t>(data);
      if (!has_offsets) {
         cursor = data + size;
      }
      if (!out.empty() && out.back().key >= container.key) {
         return bad("container keys are not ascending");
      }
      out.push_back(container);
   }
   return SILO_GPU_OK;
}

/// One block per roaring container: its values become bits of row[key * 1024 ...] (OR-ed in; ids >= n_rows are ignored).
__global__ __launch_bounds__(256) void k_expand_roaring(
   const uint8_t* __restrict__ payload, const RoaringContainer* __restrict__ containers, uint64_t* __restrict__ row, uint32_t row_words, uint32_t n_rows
) {
   const RoaringContainer container = containers[blockIdx.x];
   const uint8_t* bytes = payload + container.offset;  // byte loads: the format aligns nothing
   const auto data = [&](uint32_t index) { return static_cast<uint32_t>(bytes[2 * index]) | (static_cast<uint32_t>(bytes[2 * index + 1]) << 8); };
   const uint64_t base = static_cast<uint64_t>(container.key) << 16;
   const auto setBits = [&](uint32_t first, uint32_t last) {  // values first..last of this container, inclusive
      for (uint32_t word = first >> 6; word <= (last >> 6); ++word) {
         const uint32_t lo = word == (first >> 6) ? (first & 63u) : 0u;
         const uint32_t hi = word == (last >> 6) ? (last & 63u) : 63u;
         uint64_t mask = (hi == 63u ? ~0ull : ((1ull << (hi + 1)) - 1ull)) & ~((1ull << lo) - 1ull);
         const uint64_t row_word = (base >> 6) + word;
         const uint64_t first_id = row_word * 64u;
         if (row_word >= row_words || first_id >= n_rows) {
            return;
         }
         if (first_id + 64u > n_rows) {
            mask &= (1ull << (n_rows - first_id)) - 1ull;
         }
         atomicOr(reinterpret_cast<unsigned long long*>(row + row_word), static_cast<unsigned long long>(mask));
      }
   };
   if (container.type == ROARING_BITSET) {
      for (uint32_t word = threadIdx.x; word < 1024; word += blockDim.x) {
         uint64_t value = 0;
         for (uint32_t part = 0; part < 4; ++part) {
            value |= static_cast<uint64_t>(data(word * 4 + part)) << (16 * part);
         }
         const uint64_t row_word = (base >> 6) + word;
         const uint64_t first_id = row_word * 64u;
         if (value != 0 && row_word < row_words && first_id < n_rows) {
            if (first_id + 64u > n_rows) {
               value &= (1ull << (n_rows - first_id)) - 1ull;
            }
            atomicOr(reinterpret_cast<unsigned long long*>(row + row_word), static_cast<unsigned long long>(value));
         }
      }
   } else if (container.type == ROARING_ARRAY) {
      for (uint32_t i = threadIdx.x; i < container.count; i += blockDim.x) {
         setBits(data(i), data(i));
      }
   } else {
      for (uint32_t i = threadIdx.x; i < container.count; i += blockDim.x) {
         const uint32_t start = data(2 * i);
         const uint32_t last = min(65535u, start + data(2 * i + 1));
         setBits(start, last);
      }
   }
}

/// row = valid & ~row  (a flipped bitmap, position.cpp:70-100), or row = valid & ~(a | b) for the deleted symbol.
__global__ void k_complement_rows(uint64_t* __restrict__ out, const uint64_t* __restrict__ a, const uint64_t* __restrict__ b, uint32_t row_words, uint32_t n_rows) {
   const uint32_t w = blockIdx.x * blockDim.x + threadIdx.x;
   if (w < row_words) {
      out[w] = silo_gpu::valid_mask(w, n_rows) & ~(a[w] | (b != nullptr ? b[w] : 0ull));
   }
}

/// Merges the one-hot row of `symbol` at `position` into the build-time planes, where append_sequences would have put it.
__global__ __launch_bounds__(256) void k_merge_symbol_row(
   const SeqStoreDev store, uint32_t position, uint32_t symbol, const uint64_t* __restrict__ row, uint64_t* __restrict__ seen, uint64_t* sparse,
   uint32_t* sparse_count, uint32_t sparse_capacity
) {
   const uint32_t w = blockIdx.x * blockDim.x + threadIdx.x;
   if (w >= store.row_words) {
      return;
   }
   const uint64_t bits = row[w];
   seen[w] |= bits;
   if (bits == 0) {
      return;
   }
   const uint8_t kind = store.kind[symbol];
   if (kind == PLANE_SCAN) {
      const uint32_t code = static_cast<uint32_t>(store.index[symbol]) + 1u;
      uint64_t* planes = store.scan + static_cast<size_t>(position) * store.n_bits * store.row_words + w;
      for (uint32_t bit = 0; bit < store.n_bits; ++bit) {
         if (((code >> bit) & 1u) != 0) {
            planes[static_cast<size_t>(bit) * store.row_words] |= bits;
         }
      }
   } else if (kind == PLANE_EXTRA) {
      planePtr(store, position, symbol)[w] |= bits;
   } else {
      for (uint64_t open = bits; open != 0; open &= open - 1) {
         const uint32_t slot = atomicAdd(sparse_count, 1u);
         if (slot < sparse_capacity) {
            sparse[slot] = (static_cast<uint64_t>(position) << 37) | (static_cast<uint64_t>(symbol) << 32) |
                           (static_cast<uint64_t>(w) * 64u + static_cast<uint32_t>(__builtin_ctzll(open)));
         }
      }
   }
}

/// (position, row) pairs of the missing symbol -> bits of its column planes.
__global__ void k_scatter_missing(const SeqStoreDev store, const uint32_t* __restrict__ positions, const uint32_t* __restrict__ rows, uint32_t n_pairs) {
   const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
   if (i < n_pairs && positions[i] < store.positions) {
      uint64_t* plane = planePtr(store, positions[i], store.missing_symbol);
      atomicOr(reinterpret_cast<unsigned long long*>(plane + (rows[i] >> 6)), 1ull << (rows[i] & 63u));
   }
}

/// Expands one payload into store->d_import_row (zeroed first).
int expandPayload(silo_gpu_store* store, const void* bytes, size_t n_bytes) {
   const size_t row_bytes = static_cast<size_t>(store->row_words) * sizeof(uint64_t);
   HIP_TRY(hipMemsetAsync(store->d_import_row, 0, row_bytes, nullptr));
   std::vector<RoaringContainer> containers;
   if (const int rc = parseRoaringDirectory(static_cast<const uint8_t*>(bytes), n_bytes, containers); rc != SILO_GPU_OK) {
      return rc;
   }
   if (containers.empty()) {
      return SILO_GPU_OK;
   }
   uint8_t* d_payload = nullptr;
   RoaringContainer* d_containers = nullptr;
   HIP_TRY(hipMalloc(&d_payload, (n_bytes + 7) / 8 * 8));
   hipError_t status = hipMalloc(&d_containers, containers.size() * sizeof(RoaringContainer));
   if (status == hipSuccess) {
      status = hipMemcpy(d_payload, bytes, n_bytes, hipMemcpyHostToDevice);  // ONE contiguous copy
   }
   if (status == hipSuccess) {
      status = hipMemcpy(d_containers, containers.data(), containers.size() * sizeof(RoaringContainer), hipMemcpyHostToDevice);
   }
   if (status == hipSuccess) {
      k_expand_roaring<<<static_cast<uint32_t>(containers.size()), 256>>>(d_payload, d_containers, store->d_import_row, store->row_words, store->sequence_count);
      status = hipDeviceSynchronize();
   }
   (void)hipFree(d_payload);
   (void)hipFree(d_containers);
   HIP_TRY(status);
   return SILO_GPU_OK;
}

}  // namespace

extern "C" {

int silo_gpu_store_import_missing_rows(
   silo_gpu_store* store, uint32_t seqstore_id, uint32_t first_sequence, uint32_t n_sequences, const silo_gpu_roaring_payload* rows
) {
   if (store == nullptr || seqstore_id >= store->seqstores.size() || (rows == nullptr && n_sequences != 0) ||
       static_cast<uint64_t>(first_sequence) + n_sequences > store->sequence_count) {
      return fail(SILO_GPU_ERR_INVALID_ARGUMENT, "silo_gpu_store_import_missing_rows: bad arguments");
   }
   std::lock_guard<std::mutex> lock(store->mutex);
   HIP_TRY(hipSetDevice(store->device));
   SeqStoreHost& seqstore = store->seqstores[seqstore_id];
   if (const int rc = ensureBuildPlanes(store, seqstore); rc != SILO_GPU_OK) {
      return rc;
   }
   if (seqstore.dev.kind[seqstore.dev.missing_symbol] != PLANE_EXTRA || seqstore.dev.build_mode != BUILD_PLANES) {
      return fail(SILO_GPU_ERR_UNSUPPORTED, "silo_gpu_store_import_missing_rows: the missing symbol has no column plane in this store (or the store is being built in two passes)");
   }
   seqstore.totals_ready = false;
   // the row-wise bitmaps hold POSITIONS (a few runs per row): their directory and values are read on the host
   std::vector<uint32_t> pair_positions;
   std::vector<uint32_t> pair_rows;
   std::vector<RoaringContainer> containers;
   for (uint32_t r = 0; r < n_sequences; ++r) {
      const auto* bytes = static_cast<const uint8_t*>(rows[r].bytes);
      if (const int rc = parseRoaringDirectory(bytes, rows[r].n_bytes, containers); rc != SILO_GPU_OK) {
         return rc;
      }
      for (const RoaringContainer& container : containers) {
         const auto u16 = [&](size_t index) { return static_cast<uint32_t>(bytes[container.offset + 2 * index]) | (static_cast<uint32_t>(bytes[container.offset + 2 * index + 1]) << 8); };
         const uint32_t base = container.key << 16;
         const auto add = [&](uint32_t position) {
            if (position < seqstore.dev.positions) {
               pair_positions.push_back(position);
               pair_rows.push_back(first_sequence + r);
            }
         };
         if (container.type == ROARING_ARRAY) {
            for (uint32_t i = 0; i < container.count; ++i) {
               add(base + u16(i));
            }
         } else if (container.type == ROARING_RUN) {
            for (uint32_t i = 0; i < container.count; ++i) {
               const uint32_t start = u16(2 * i);
               for (uint32_t value = start; value <= std::min(65535u, start + u16(2 * i + 1)); ++value) {
                  add(base + value);
               }
            }
         } else {
            for (uint32_t value = 0; value < 65536; ++value) {
               if ((bytes[container.offset + value / 8] >> (value % 8)) & 1u) {
                  add(base + value);
               }
            }
         }
      }
   }
   if (pair_rows.empty()) {
      return SILO_GPU_OK;
   }
   uint32_t* d_pairs = nullptr;
   const size_t n = pair_rows.size();
   HIP_TRY(hipMalloc(&d_pairs, 2 * n * sizeof(uint32_t)));
   hipError_t status = hipMemcpy(d_pairs, pair_positions.data(), n * sizeof(uint32_t), hipMemcpyHostToDevice);
   if (status == hipSuccess) {
      status = hipMemcpy(d_pairs + n, pair_rows.data(), n * sizeof(uint32_t), hipMemcpyHostToDevice);
   }
   if (status == hipSuccess) {
      k_scatter_missing<<<static_cast<uint32_t>((n + 255) / 256), 256>>>(seqstore.dev, d_pairs, d_pairs + n, static_cast<uint32_t>(n));
      status = hipDeviceSynchronize();
   }
   (void)hipFree(d_pairs);
   HIP_TRY(status);
   return SILO_GPU_OK;
}

int silo_gpu_store_import_position(
   silo_gpu_store* store, uint32_t seqstore_id, uint32_t position, const silo_gpu_roaring_payload* bitmaps, uint32_t n_bitmaps,
   uint32_t flipped_symbol, uint32_t deleted_symbol
) {
   if (store == nullptr || seqstore_id >= store->seqstores.size() || (bitmaps == nullptr && n_bitmaps != 0)) {
      return fail(SILO_GPU_ERR_INVALID_ARGUMENT, "silo_gpu_store_import_position: bad arguments");
   }
   std::lock_guard<std::mutex> lock(store->mutex);
   HIP_TRY(hipSetDevice(store->device));
   SeqStoreHost& seqstore = store->seqstores[seqstore_id];
   const SeqStoreDev& dev = seqstore.dev;
   const auto valid_symbol = [&](uint32_t symbol) { return symbol == SILO_GPU_SYMBOL_NONE || (symbol < dev.n_symbols && symbol != dev.missing_symbol); };
   if (position >= dev.positions || !valid_symbol(flipped_symbol) || !valid_symbol(deleted_symbol)) {
      return fail(SILO_GPU_ERR_INVALID_ARGUMENT, "silo_gpu_store_import_position: position or flipped / deleted symbol out of range");
   }
   if (seqstore.dev.build_mode != BUILD_PLANES) {
      return fail(SILO_GPU_ERR_UNSUPPORTED, "silo_gpu_store_import_position: the store is being built in two passes");
   }
   if (const int rc = ensureBuildPlanes(store, seqstore); rc != SILO_GPU_OK) {
      return rc;
   }
   seqstore.finalized = false;
   seqstore.totals_ready = false;
   const size_t row_bytes = static_cast<size_t>(store->row_words) * sizeof(uint64_t);
   if (store->d_import_row == nullptr) {
      HIP_TRY(hipMalloc(&store->d_import_row, row_bytes));
      HIP_TRY(hipMalloc(&store->d_import_union, row_bytes));
   }
   HIP_TRY(hipMemsetAsync(store->d_import_union, 0, row_bytes, nullptr));
   const uint32_t blocks = (store->row_words + 255) / 256;
   uint32_t count_before = 0;
   HIP_TRY(hipMemcpy(&count_before, seqstore.d_sparse_count, sizeof(uint32_t), hipMemcpyDeviceToHost));
   const auto merge = [&](uint32_t symbol) -> int {
      // a sparsely stored symbol appends one key per row: make room for every row of the store (an import is not a hot path)
      if (dev.kind[symbol] == PLANE_SPARSE) {
         if (const int rc = growSparse(seqstore, count_before + store->sequence_count); rc != SILO_GPU_OK) {
            return rc;
         }
      }
      k_merge_symbol_row<<<blocks, 256>>>(
         seqstore.dev, position, symbol, store->d_import_row, store->d_import_union, seqstore.d_sparse, seqstore.d_sparse_count, seqstore.sparse_capacity
      );
      HIP_TRY(hipDeviceSynchronize());
      HIP_TRY(hipMemcpy(&count_before, seqstore.d_sparse_count, sizeof(uint32_t), hipMemcpyDeviceToHost));
      return SILO_GPU_OK;
   };
   for (uint32_t k = 0; k < n_bitmaps; ++k) {
      const uint32_t symbol = bitmaps[k].symbol;
      if (symbol >= dev.n_symbols || symbol == dev.missing_symbol) {
         return fail(SILO_GPU_ERR_INVALID_ARGUMENT, "silo_gpu_store_import_position: symbol out of range (the missing symbol is imported row-wise)");
      }
      if (symbol == deleted_symbol) {
         continue;  // stored empty: rebuilt below from what is left
      }
      if (const int rc = expandPayload(store, bitmaps[k].bytes, bitmaps[k].n_bytes); rc != SILO_GPU_OK) {
         return rc;
      }
      if (symbol == flipped_symbol) {  // stored as its complement (position.cpp:70-100)
         k_complement_rows<<<blocks, 256>>>(store->d_import_row, store->d_import_row, nullptr, store->row_words, store->sequence_count);
         HIP_TRY(hipGetLastError());
      }
      if (const int rc = merge(symbol); rc != SILO_GPU_OK) {
         return rc;
      }
   }
   if (deleted_symbol != SILO_GPU_SYMBOL_NONE) {
      // the deleted symbol's rows: every row no other symbol claims and that is not missing here (position.cpp:102-127)
      const uint64_t* missing = dev.kind[dev.missing_symbol] == PLANE_EXTRA ? planePtr(dev, position, dev.missing_symbol) : nullptr;
      k_complement_rows<<<blocks, 256>>>(store->d_import_row, store->d_import_union, missing, store->row_words, store->sequence_count);
      HIP_TRY(hipGetLastError());
      if (const int rc = merge(deleted_symbol); rc != SILO_GPU_OK) {
         return rc;
      }
   }
   return SILO_GPU_OK;
}

}  // extern "C"
