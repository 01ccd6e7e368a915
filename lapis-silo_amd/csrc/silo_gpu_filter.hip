// silo_gpu_filter.hip — K2 k_popcount, K3 k_filter_eval, K3b k_filter_eval_batch (the fused operator tree of
// operators/{index_scan,complement,intersection,union,threshold,full,empty,bitmap_selection}.cpp), count slots, row bitsets
// from metadata, the plane of a single symbol at a position (filter leaves) and FastaAligned.
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
#include <type_traits>
#include <string>
#include <vector>

#include "store_internal.h"

using namespace silo_gpu_detail;

namespace {

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
   unsigned long long* host_parts;   // count slot: page-locked host words, one per block: epoch << 32 | rows the block selected
   uint32_t epoch;                   // count slot: of this launch
   uint32_t n_tiles;                 // 128-word tiles of the row (k_filter_eval_parts)
   const uint64_t* leaves[SILO_GPU_MAX_LEAVES];
   uint32_t code[2 * SILO_GPU_MAX_INSTRUCTIONS];
};

// Tried and dropped (profiles/r01_k3_variants.md): 16 instead of 8 leaf loads in flight per n-ary instruction, and
// fetching the first 24 leaves up front into LDS (register-staged: spilled to scratch; LDS-DMA global_load_lds_dwordx4:
// no spill) — neither moved the kernel time of the 32-column program (19-21 us at 10 M sequences either way).  Round 3:
// blocks of 4 waves that fetch ALL leaves of a 128-word tile into LDS at once (32 loads in flight per tile) before wave 0
// evaluates: 30 us instead of 20 (profiles/r03_notes.md) — the kernel is not waiting for its loads.
/// The end of a filter kernel's wave (64 lanes, one per pair of result words): the popcount of the result goes to the count shards.
__device__ __forceinline__ void deliverFilterCount(const FilterEvalArgs& args, silo_gpu::Word2 result) {
   if (args.out_count != nullptr) {
      const uint32_t bits = static_cast<uint32_t>(__popcll(result.x)) + static_cast<uint32_t>(__popcll(result.y));
      addToCountShard(args.out_count, waveSumToLane63(bits));
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
   deliverFilterCount(args, result);
}

/// The filter of ONE query whose cardinality the host waits for (count slot).  Round 2 found the launch's last block on the
/// device — a returning atomic on a count shard, a ticket per shard class, a main ticket, the sum of the shards, then the
/// store to the host: five dependent round trips to the memory side (device-scope atomics are performed there, the L2s of the
/// 8 XCDs not being coherent with each other), ~10 us of a 20 us kernel.  Here a block hands ITS count straight to the
/// host — one posted 8-byte store into page-locked memory, tagged with the launch's epoch — and the host adds the parts up
/// as they arrive: no atomic, no ticket, nothing to re-arm.  Blocks of several waves, every wave on its own tiles of the
/// row (slots in LDS per wave, no barrier but the one before the block's store), at most COUNT_MAX_PARTS blocks.
constexpr uint32_t COUNT_MAX_PARTS = 2048;
/// WAVES per block; WORDS per lane: 2 (16-byte accesses) or 1.  One word per lane — twice the waves for a row — was tried
/// for the query on its own (10 M rows are only 1 221 waves of 2 words on 1 024 SIMDs): no difference, 29.7 against 29.0 us
/// end to end for configs[2] (profiles/r03_notes.md); the launches use 2.
template <uint32_t BATCH, uint32_t WAVES, uint32_t WORDS>
__global__ __launch_bounds__(WAVES * 64) void k_filter_eval_parts(const FilterEvalArgs args) {
   using silo_gpu::Word2;
   using Word = std::conditional_t<WORDS == 2, Word2, uint64_t>;
   using Stored = std::conditional_t<WORDS == 2, ulonglong2, uint64_t>;
   extern __shared__ ulonglong2 s_slots[];  // [wave][n_slots][64] of Stored
   __shared__ uint32_t s_part[WAVES];
   const uint32_t lane = threadIdx.x & 63u;
   const uint32_t wave = threadIdx.x >> 6;
   Stored* slots = reinterpret_cast<Stored*>(s_slots) + static_cast<size_t>(wave) * args.n_slots * 64u;
   uint32_t selected = 0;
   for (uint32_t tile = blockIdx.x * WAVES + wave; tile < args.n_tiles; tile += gridDim.x * WAVES) {  // (uniform per wave)
      const uint32_t w = (tile * 64u + lane) * WORDS;  // row_words is a multiple of 32
      const bool active = w < args.row_words;
      const uint32_t w_safe = active ? w : 0;
      Word valid = silo_gpu::zeroOf<Word>();
      if (active) {
         if constexpr (WORDS == 2) {
            valid = {silo_gpu::valid_mask(w, args.sequence_count), silo_gpu::valid_mask(w + 1, args.sequence_count)};
         } else {
            valid = silo_gpu::valid_mask(w, args.sequence_count);
         }
      }
      const auto leaf = [&](uint32_t index) -> Word {
         if constexpr (WORDS == 2) {
            const ulonglong2 v = *reinterpret_cast<const ulonglong2*>(args.leaves[index] + w_safe);
            return {v.x, v.y};
         } else {
            return args.leaves[index][w_safe];
         }
      };
      const auto get = [&](uint32_t index) -> Word {
         if (index >= SILO_GPU_LEAF_OPERAND) {
            return leaf(index - SILO_GPU_LEAF_OPERAND);
         }
         if constexpr (WORDS == 2) {
            const ulonglong2 v = slots[index * 64u + lane];
            return {v.x, v.y};
         } else {
            return slots[index * 64u + lane];
         }
      };
      const auto set = [&](uint32_t index, Word value) {
         if constexpr (WORDS == 2) {
            slots[index * 64u + lane] = make_ulonglong2(value.x, value.y);
         } else {
            slots[index * 64u + lane] = value;
         }
      };
      Word result = silo_gpu::bitprog_run<Word, BATCH>(args.code, args.n_instructions, valid, get, set, leaf);
      result = result & valid;
      if constexpr (WORDS == 2) {
         if (active && args.out != nullptr) {
            *reinterpret_cast<ulonglong2*>(args.out + w) = make_ulonglong2(result.x, result.y);
         }
         selected += static_cast<uint32_t>(__popcll(result.x)) + static_cast<uint32_t>(__popcll(result.y));
      } else {
         if (active && args.out != nullptr) {
            args.out[w] = result;
         }
         selected += static_cast<uint32_t>(__popcll(result));
      }
   }
   selected = waveSumToLane63(selected);
   if (lane == 63u) {
      s_part[wave] = selected;
   }
   __syncthreads();
   if (threadIdx.x == 0) {
      uint32_t total = 0;
#pragma unroll
      for (uint32_t k = 0; k < WAVES; ++k) {
         total += s_part[k];
      }
      __hip_atomic_store(
         args.host_parts + blockIdx.x, (static_cast<unsigned long long>(args.epoch) << 32) | total, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM
      );
   }
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


}  // namespace

extern "C" {

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
   unsigned long long* host_parts = nullptr;      // page-locked, COUNT_MAX_PARTS words: epoch << 32 | the block's count
   unsigned long long* host_parts_dev = nullptr;  // their device address
   uint32_t epoch = 0;                            // of the launch in flight (never 0: the words start out as 0)
   uint32_t n_parts = 0;                          // blocks of the launch in flight
};

namespace {
int filterEvalLaunch(
   const silo_gpu_store* store, const silo_gpu_bitprog* program, uint64_t* out_bitset_dev, uint64_t* out_count_dev, silo_gpu_count_slot* slot, void* stream
);
}  // namespace

int silo_gpu_filter_eval(const silo_gpu_store* store, const silo_gpu_bitprog* program, uint64_t* out_bitset_dev, uint64_t* out_count_dev, void* stream) {
   return filterEvalLaunch(store, program, out_bitset_dev, out_count_dev, nullptr, stream);
}

int silo_gpu_count_slot_create(silo_gpu_count_slot** out_slot) {
   if (out_slot == nullptr) {
      return fail(SILO_GPU_ERR_INVALID_ARGUMENT, "silo_gpu_count_slot_create: null out pointer");
   }
   auto* slot = new (std::nothrow) silo_gpu_count_slot;
   if (slot == nullptr) {
      return fail(SILO_GPU_ERR_OUT_OF_MEMORY, "out of host memory");
   }
   hipError_t err = hipHostMalloc(&slot->host_parts, COUNT_MAX_PARTS * sizeof(unsigned long long), hipHostMallocMapped | hipHostMallocCoherent);
   if (err == hipSuccess) {
      memset(slot->host_parts, 0, COUNT_MAX_PARTS * sizeof(unsigned long long));
      err = hipHostGetDevicePointer(reinterpret_cast<void**>(&slot->host_parts_dev), slot->host_parts, 0);
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
      if (slot->host_parts != nullptr) {
         (void)hipHostFree(slot->host_parts);
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
   slot->epoch = slot->epoch == 0xFFFFFFFFu ? 1u : slot->epoch + 1u;  // a part counts only when it carries this launch's epoch
   slot->n_parts = 0;
   return filterEvalLaunch(store, program, out_bitset_dev, nullptr, slot, stream);
}

int silo_gpu_count_slot_wait(silo_gpu_count_slot* slot, uint64_t* out_count, void* stream) {
   if (slot == nullptr || out_count == nullptr) {
      return fail(SILO_GPU_ERR_INVALID_ARGUMENT, "silo_gpu_count_slot_wait: null argument");
   }
   // Every block stores its part with system scope.  The wait is a pure spin on those words — no HIP call from the polling
   // threads (round 1 polled hipStreamQuery from every request thread; under rocprofv3's kernel tracing that run segfaulted,
   // and whether the fault was the profiler's or the polling's was never established, so the polling is gone).  A launch that
   // does not deliver within the spin budget (tens of milliseconds: a failed or wedged launch, or a very busy device) is
   // waited for with ONE blocking hipStreamSynchronize, which also reports a broken stream.
   constexpr uint64_t SPIN_BUDGET = uint64_t{1} << 22;
   uint64_t spins = 0;
   uint64_t total = 0;
   bool synchronised = false;
   for (uint32_t part = 0; part < slot->n_parts; ++part) {
      for (;;) {
         const unsigned long long value = __atomic_load_n(slot->host_parts + part, __ATOMIC_ACQUIRE);
         if (static_cast<uint32_t>(value >> 32) == slot->epoch) {
            total += static_cast<uint32_t>(value);
            break;
         }
         if (synchronised) {
            return fail(SILO_GPU_ERR_HIP, "count slot: the kernel finished without delivering its total");
         }
         if (++spins >= SPIN_BUDGET) {
            HIP_TRY(hipStreamSynchronize(static_cast<hipStream_t>(stream)));
            synchronised = true;
            continue;
         }
#if defined(__x86_64__)
         __builtin_ia32_pause();
#endif
      }
   }
   *out_count = total;
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
   const silo_gpu_store* store, const silo_gpu_bitprog* program, uint64_t* out_bitset_dev, uint64_t* out_count_dev, silo_gpu_count_slot* slot, void* stream
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
   for (uint32_t k = 0; k < program->n_leaves; ++k) {
      args.leaves[k] = program->leaves[k];
   }
   memcpy(args.code, program->code, static_cast<size_t>(program->n_instructions) * 2 * sizeof(uint32_t));
   const bool wide = g_tune_eval_leaf_batch.load() == 16;
   if (slot != nullptr) {  // the host waits for the cardinality: a part per block, straight into page-locked memory
      // Waves per block: 8 while the slots of 8 waves fit the LDS, else 4.  Fewer, fatter blocks = fewer parts: the posted
      // 8-byte stores to the host are the scarce thing when several clients query at once (8 one-by-one clients with a part per
      // 4 waves: 306 stores per query, 23 M/s, and throughput fell to 76 k queries/s — profiles/r03_notes.md).
      const bool fat = program->n_slots <= 16;
      const uint32_t waves = fat ? 8u : 4u;
      args.n_tiles = (store->row_words + 127u) / 128u;
      const uint32_t blocks = std::min<uint32_t>((args.n_tiles + waves - 1) / waves, COUNT_MAX_PARTS);
      args.host_parts = slot->host_parts_dev;
      args.epoch = slot->epoch;
      const size_t lds_bytes = static_cast<size_t>(program->n_slots) * waves * 64 * sizeof(ulonglong2);  // <= 128 KiB
      static std::once_flag lds_once;
      std::call_once(lds_once, [] {
         const int most = 128 * 1024;
         (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_filter_eval_parts<8, 4, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, most);
         (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_filter_eval_parts<16, 4, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, most);
         (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_filter_eval_parts<8, 8, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, most);
         (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_filter_eval_parts<16, 8, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, most);
      });
      const auto hip_stream = static_cast<hipStream_t>(stream);
      if (fat) {
         if (wide) {
            k_filter_eval_parts<16, 8, 2><<<blocks, 8 * 64, lds_bytes, hip_stream>>>(args);
         } else {
            k_filter_eval_parts<8, 8, 2><<<blocks, 8 * 64, lds_bytes, hip_stream>>>(args);
         }
      } else if (wide) {
         k_filter_eval_parts<16, 4, 2><<<blocks, 4 * 64, lds_bytes, hip_stream>>>(args);
      } else {
         k_filter_eval_parts<8, 4, 2><<<blocks, 4 * 64, lds_bytes, hip_stream>>>(args);
      }
      HIP_TRY(hipGetLastError());
      slot->n_parts = blocks;
      return SILO_GPU_OK;
   }
   const uint32_t blocks = (store->row_words + EVAL_WORDS_PER_BLOCK - 1) / EVAL_WORDS_PER_BLOCK;
   const size_t lds_bytes = static_cast<size_t>(program->n_slots) * EVAL_THREADS * sizeof(ulonglong2);  // <= 32 KiB
   if (wide) {
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


}  // extern "C"
