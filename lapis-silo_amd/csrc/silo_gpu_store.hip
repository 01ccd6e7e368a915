// silo_gpu_store.hip — a store's lifetime and build: B1 k_transpose_sequences / B2 k_generate_synthetic (aligned sequences ->
// planes, storage/sequence_store.cpp:100-190), the runs of the missing symbol, and finalize: the layout of every position
// (layout_choice.h), the adaptive planes, the escape keys in both orders.  DESIGN.md sections 2 and 3.
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

#include "store_internal.h"

using namespace silo_gpu_detail;

namespace {

int buildLayout(silo_gpu_store* store, SeqStoreHost& seqstore);  // the adaptive code planes, defined next to the scan launchers
bool reencodes(const silo_gpu_store* store, const SeqStoreDev& dev);
int planLayout(const silo_gpu_store* store, SeqStoreHost& seqstore, SeqStoreHost::LayoutWork& work, bool zero_planes, bool allow_implicit, bool* fits);


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

/// The slice-major keys as k_scan_escapes_sliced reads them (SeqStoreHost::Layout::d_escapes_sliced): one block per granule of
/// ESCAPE_GRANULE_KEYS keys of one slice.  `unpacked_first[g]` = index of the granule's first key in the sorted 8-byte list,
/// `unpacked_end[g]` = one past its last (a slice's last granule is short: the rest is padding).
__global__ __launch_bounds__(256) void k_pack_sliced_keys(
   const uint64_t* __restrict__ keys, const uint32_t* __restrict__ unpacked_first, const uint32_t* __restrict__ unpacked_end, uint32_t n_scan,
   uint32_t* __restrict__ packed, uint32_t* __restrict__ granule_base, uint64_t* __restrict__ overflow, uint32_t* __restrict__ overflow_count,
   uint32_t overflow_capacity
) {
   const uint32_t granule = blockIdx.x;
   const uint32_t begin = unpacked_first[granule];
   const uint32_t end = unpacked_end[granule];
   const auto counter_of = [n_scan](uint64_t key) { return static_cast<uint32_t>(key >> 37) * n_scan + (static_cast<uint32_t>(key >> 32) & 31u); };
   const uint32_t base = begin < end ? counter_of(keys[begin]) : 0u;
   if (threadIdx.x == 0) {
      granule_base[granule] = base;
   }
   for (uint32_t k = threadIdx.x; k < ESCAPE_GRANULE_KEYS; k += blockDim.x) {
      uint32_t value = ESCAPE_KEY_INVALID;
      if (begin + k < end) {
         const uint64_t key = keys[begin + k];
         const uint32_t relative = counter_of(key) - base;
         if (relative <= ESCAPE_MAX_RELATIVE) {
            value = (relative << ESCAPE_SLICE_SHIFT) | (static_cast<uint32_t>(key) & ESCAPE_ROW_MASK);
         } else {
            const uint32_t slot = atomicAdd(overflow_count, 1u);
            if (slot < overflow_capacity) {
               overflow[slot] = (static_cast<uint64_t>(counter_of(key)) << 32) | (key & 0xFFFFFFFFull);
            }
         }
      }
      packed[static_cast<size_t>(granule) * ESCAPE_GRANULE_KEYS + k] = value;
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


}  // namespace

namespace silo_gpu_detail {

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

}  // namespace silo_gpu_detail

extern "C" {

int silo_gpu_store_create(const silo_gpu_store_desc* desc, silo_gpu_store** out) {
   if (desc == nullptr || out == nullptr || desc->n_seqstores == 0 || desc->seqstores == nullptr) {
      return fail(SILO_GPU_ERR_INVALID_ARGUMENT, "silo_gpu_store_create: null descriptor");
   }
   *out = nullptr;
   if (int rc = ensureDevice(desc->device); rc != SILO_GPU_OK) {
      return rc;
   }
   {
      // ONE device per process (one process per GPU, as the engine is deployed): the per-thread side streams, the scratch
      // pools and the kernels' shared-memory attributes are set up once, for the device of the first store
      static std::atomic<int> process_device{-1};
      int expected = -1;
      if (!process_device.compare_exchange_strong(expected, desc->device) && expected != desc->device) {
         return fail(
            SILO_GPU_ERR_UNSUPPORTED, "silo_gpu_store_create: this process holds stores on device " + std::to_string(expected) +
                                         " already; one process serves one GPU (start a process per device)"
         );
      }
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
      (void)hipFree(seqstore.layout.d_granule_base);
      (void)hipFree(seqstore.layout.d_escapes_overflow);
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


}  // extern "C"

namespace {

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
   // the slice-major copy of the keys for the scan's escape pass — packed to 4 bytes per key — and where each position's keys
   // begin in every slice
   uint32_t* d_escapes_sliced = nullptr;
   uint32_t* d_granule_base = nullptr;
   uint64_t* d_escapes_overflow = nullptr;
   uint32_t n_overflow = 0;
   uint64_t packed_keys = 0;
   uint32_t* d_slice_first = nullptr;
   std::vector<uint32_t> slice_first;
   const uint32_t n_slices = (store->sequence_count + (1u << ESCAPE_SLICE_SHIFT) - 1) >> ESCAPE_SLICE_SHIFT;
   if (work.total_escapes > 0 && n_slices <= ESCAPE_MAX_SLICES) {
      const size_t n_entries = static_cast<size_t>(n_slices) * (positions + 1);
      uint64_t* d_sorted = nullptr;  // the keys slice-major, 8 bytes wide: what the packed list is made from
      uint32_t* d_unpacked = nullptr;
      uint32_t* d_overflow_count = nullptr;
      const auto discardSliced = [&]() {
         (void)hipFree(d_sorted);
         (void)hipFree(d_unpacked);
         (void)hipFree(d_overflow_count);
         (void)hipFree(d_escapes_sliced);
         (void)hipFree(d_granule_base);
         (void)hipFree(d_escapes_overflow);
         (void)hipFree(d_slice_first);
      };
      hipError_t status = hipMalloc(&d_sorted, work.escape_bytes);
      status = status != hipSuccess ? status : hipMalloc(&d_slice_first, n_entries * sizeof(uint32_t));
      status = status != hipSuccess ? status : hipMemcpy(d_sorted, work.d_escapes, work.total_escapes * sizeof(uint64_t), hipMemcpyDeviceToDevice);
      if (status != hipSuccess) {
         discardSliced();
         SILO_LAYOUT_TRY(status);
      }
      if (const int rc = silo_gpu_internal_sort_keys_by_bits(d_sorted, work.total_escapes, ESCAPE_SLICE_SHIFT, ESCAPE_SLICE_SHIFT + ESCAPE_SLICE_BITS); rc != SILO_GPU_OK) {
         discardSliced();
         work.discard();
         return rc;
      }
      k_slice_index<<<static_cast<uint32_t>((n_entries + 255) / 256), 256>>>(
         d_sorted, static_cast<uint32_t>(work.total_escapes), ESCAPE_SLICE_SHIFT, n_slices, positions, d_slice_first
      );
      slice_first.resize(n_entries);
      status = hipGetLastError();
      status = status != hipSuccess ? status : hipMemcpy(slice_first.data(), d_slice_first, n_entries * sizeof(uint32_t), hipMemcpyDeviceToHost);
      if (status != hipSuccess) {
         discardSliced();
         SILO_LAYOUT_TRY(status);
      }
      // every slice's keys padded to whole granules; the index [slice][position] moves to the packed numbering
      std::vector<uint32_t> unpacked_first, unpacked_end;
      for (uint32_t slice = 0; slice < n_slices; ++slice) {
         uint32_t* first = slice_first.data() + static_cast<size_t>(slice) * (positions + 1);
         const uint32_t slice_begin = first[0];
         const uint32_t slice_end = first[positions];
         const auto packed_begin = static_cast<uint32_t>(unpacked_first.size()) * ESCAPE_GRANULE_KEYS;
         for (uint32_t at = slice_begin; at < slice_end; at += ESCAPE_GRANULE_KEYS) {
            unpacked_first.push_back(at);
            unpacked_end.push_back(std::min(slice_end, at + ESCAPE_GRANULE_KEYS));
         }
         for (uint32_t p = 0; p <= positions; ++p) {
            first[p] = packed_begin + (first[p] - slice_begin);
         }
      }
      const size_t n_granules = unpacked_first.size();
      packed_keys = static_cast<uint64_t>(n_granules) * ESCAPE_GRANULE_KEYS;
      const uint32_t overflow_capacity = static_cast<uint32_t>(std::min<uint64_t>(work.total_escapes, uint64_t{1} << 26));
      if (packed_keys >= (uint64_t{1} << 32)) {
         discardSliced();
         work.discard();
         return fail(SILO_GPU_ERR_UNSUPPORTED, "more than 2^32 packed escape keys in one sequence store");
      }
      status = hipMalloc(&d_escapes_sliced, packed_keys * sizeof(uint32_t) + 16);  // (16-byte loads: a quad of slack)
      status = status != hipSuccess ? status : hipMalloc(&d_granule_base, std::max<size_t>(n_granules, 1) * sizeof(uint32_t));
      status = status != hipSuccess ? status : hipMalloc(&d_unpacked, std::max<size_t>(n_granules, 1) * 2 * sizeof(uint32_t));
      status = status != hipSuccess ? status : hipMalloc(&d_overflow_count, sizeof(uint32_t));
      status = status != hipSuccess ? status : hipMalloc(&d_escapes_overflow, std::max<size_t>(overflow_capacity, 1) * sizeof(uint64_t));
      status = status != hipSuccess ? status : hipMemset(d_overflow_count, 0, sizeof(uint32_t));
      status = status != hipSuccess ? status : hipMemcpy(d_unpacked, unpacked_first.data(), n_granules * sizeof(uint32_t), hipMemcpyHostToDevice);
      status = status != hipSuccess ? status : hipMemcpy(d_unpacked + n_granules, unpacked_end.data(), n_granules * sizeof(uint32_t), hipMemcpyHostToDevice);
      status = status != hipSuccess ? status : hipMemcpy(d_slice_first, slice_first.data(), n_entries * sizeof(uint32_t), hipMemcpyHostToDevice);
      if (status == hipSuccess && n_granules > 0) {
         k_pack_sliced_keys<<<static_cast<uint32_t>(n_granules), 256>>>(
            d_sorted, d_unpacked, d_unpacked + n_granules, dev.n_scan, d_escapes_sliced, d_granule_base, d_escapes_overflow, d_overflow_count, overflow_capacity
         );
         status = hipGetLastError();
      }
      status = status != hipSuccess ? status : hipMemcpy(&n_overflow, d_overflow_count, sizeof(uint32_t), hipMemcpyDeviceToHost);
      if (status != hipSuccess || n_overflow > overflow_capacity) {
         discardSliced();
         SILO_LAYOUT_TRY(status);
         work.discard();
         return fail(SILO_GPU_ERR_UNSUPPORTED, "too many escape keys outside their granule's counter range");
      }
      (void)hipFree(d_sorted);
      (void)hipFree(d_unpacked);
      (void)hipFree(d_overflow_count);
      if (n_overflow == 0) {
         (void)hipFree(d_escapes_overflow);
         d_escapes_overflow = nullptr;
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
   layout.d_granule_base = d_granule_base;
   layout.d_escapes_overflow = d_escapes_overflow;
   layout.n_overflow = n_overflow;
   layout.packed_keys = packed_keys;
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
   layout.device_bytes = work.plane_bytes + work.escape_bytes + packed_keys * sizeof(uint32_t) + static_cast<size_t>(n_overflow) * sizeof(uint64_t) +
                         static_cast<size_t>(positions) * (CODE_MAP_STRIDE + 8) +
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
