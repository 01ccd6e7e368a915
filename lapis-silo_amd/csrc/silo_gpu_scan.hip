// silo_gpu_scan.hip — K1, the Mutations scan of the SILO mutation-filter hot path on CDNA4 (gfx950), and K4, its row selection.
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

#include "store_internal.h"

using namespace silo_gpu_detail;

namespace {

thread_local const char* g_last_scan_kernel = "none";

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
constexpr int SCAN_WAVES = SCAN_THREADS / 64;

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

/// One launch for up to ESCAPE_MAX_RANGES position ranges (the 12 genes of an AminoAcidMutations query): grid =
/// (blocks per slice, slice x range, filters / FILTERS); where a slice's keys of the scanned positions begin and end is read
/// from the store's slice index on the device.
struct EscapeSliceArgs {
   const uint64_t* filters[SILO_GPU_MAX_SCAN_BATCH];
   uint32_t row_words;
   uint32_t n_slices;
   uint32_t out_symbols;
   uint32_t block_keys;  // keys of a block's share: whole granules, at most ESCAPE_GRANULES_PER_BLOCK
   struct Range {
      const uint32_t* keys;          // the packed slice-major keys of the store (SeqStoreHost::Layout::d_escapes_sliced)
      const uint32_t* granule_base;  // counter of every granule's first key
      const uint32_t* slice_first;   // [n_slices][positions + 1], in the packed numbering
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
/// Keys.  4 bytes each: row within the slice | (counter - counter of the granule's first key) << 17; a granule is 4 096
/// consecutive keys of a slice, so ONE 16-byte load per lane of the block fetches a granule, four consecutive keys per lane,
/// and the granule's base counter is a scalar.
///
/// Counting.  The keys of a slice are sorted by (position, symbol), so the counters a run of keys adds to lie in a narrow
/// window behind its first key: the block counts into a window of LDS counters per filter and then adds the window to the
/// table with CONTIGUOUS atomics — 64 consecutive counters per wave instruction, the shape the memory side takes at full
/// rate; a lane per scattered counter, as the first version did, is an order of magnitude slower per add (MI355X guide,
/// "Global float atomics": access shape).  The block's share of keys is cut into chunks where the window is full: as many
/// granules as end within WINDOW counters of the chunk's first (the granules' base counters tell) — thousands of keys per
/// chunk where a position has many, one granule where private substitutions lie thirteen to a position; the window is
/// flushed and reused chunk by chunk, the filter slices stay.  Lanes whose keys share a counter add through the stretch's last
/// lane only (identical addresses do not combine for LDS atomics), and the loop body has no per-key branch (see there).  No
/// barrier between a chunk's granules: the waves run on by themselves, one waits for its keys while another counts; two
/// blocks per CU for one and two filters (<= 64 VGPRs, 64 KiB of LDS) cover each other's first and last steps.  Eight filters:
/// the slices as one byte per row and the lanes' sums in packed fields (see there), one block per CU.
constexpr uint32_t ESCAPE_GRANULES_PER_BLOCK = 64;  // of a block's share, at most
template <int FILTERS>
constexpr uint32_t escapeWindow() {  // LDS counters per filter: 48 KiB of them for 1-4 filters, 28 KiB for 8 (beside 128 KiB of filter slices)
   return FILTERS >= 8 ? 896u : 12288u / FILTERS;
}
template <int FILTERS>
constexpr uint32_t escapeLdsBytes() {
   return (FILTERS * (ESCAPE_SLICE_WORDS32 + escapeWindow<FILTERS>()) + ESCAPE_GRANULES_PER_BLOCK + 4u + 64u) * static_cast<uint32_t>(sizeof(uint32_t));
}

template <int FILTERS>
__global__ __launch_bounds__(ESCAPE_SLICE_THREADS, FILTERS <= 4 ? 8 : 4) void k_scan_escapes_sliced(const EscapeSliceArgs args, uint32_t n_filters) {
   constexpr uint32_t WINDOW = escapeWindow<FILTERS>();
   static_assert(ESCAPE_GRANULE_KEYS == ESCAPE_SLICE_THREADS * 4u, "a granule is one 16-byte load per thread of the block");
   extern __shared__ uint32_t s_filter[];  // [FILTERS][ESCAPE_SLICE_WORDS32], then the counters [FILTERS][WINDOW], the granules' bases, the chunks' last keys
   uint32_t* s_count = s_filter + FILTERS * ESCAPE_SLICE_WORDS32;
   uint32_t* s_base = s_count + FILTERS * WINDOW;  // [ESCAPE_GRANULES_PER_BLOCK + 1] the counter of every granule's first key, then one past the share's last key's
   uint32_t* s_nowhere = s_base + ESCAPE_GRANULES_PER_BLOCK + 4u;  // [64] a word per lane: where an add of nothing goes
   const uint32_t first_filter = blockIdx.z * FILTERS;
   const uint32_t slice = blockIdx.y % args.n_slices;
   const EscapeSliceArgs::Range& range = args.ranges[blockIdx.y / args.n_slices];
   const uint32_t* first = range.slice_first + static_cast<size_t>(slice) * (range.positions + 1u);
   const uint32_t key_begin = first[range.pos_begin];
   const uint32_t key_end = first[range.pos_end];
   // the block's share: args.block_keys keys (whole granules)
   const uint32_t share_begin = key_begin / ESCAPE_GRANULE_KEYS * ESCAPE_GRANULE_KEYS + blockIdx.x * args.block_keys;
   if (share_begin >= key_end) {
      return;  // (uniform) no keys for this block
   }
   const uint32_t share_end = min(share_begin + args.block_keys, key_end);
   const uint32_t range_first = range.pos_begin * args.out_symbols;
   // Everything the block reads first is asked for at once, behind the one dependent load of the slice index: the filter
   // slices, the first keys, the granules' base counters, the share's last key — every memory latency put in a row would
   // show; the keys of the granule after the next are asked for while a granule is counted, across the chunks.
   const uint32_t first_granule = share_begin / ESCAPE_GRANULE_KEYS;
   const uint32_t n_granules = (share_end - share_begin + ESCAPE_GRANULE_KEYS - 1u) / ESCAPE_GRANULE_KEYS;  // <= ESCAPE_GRANULES_PER_BLOCK
   // (unconditional: a load under a condition, or a loaded register handed on by a move, makes the compiler wait for ALL loads
   // in flight where the first is used — vmcnt(0) in the loop took a memory latency per granule: 86 us for 73 M keys.  A granule
   // past the share's last reads that one again; the whole granule exists, padded, past the slice's last key.)
   const auto loadKeys = [&](uint32_t granule) {
      const uint32_t g = min(granule, n_granules - 1u);
      const u32x4 v = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(range.keys + share_begin + g * ESCAPE_GRANULE_KEYS + threadIdx.x * 4u));
      return make_uint4(v.x, v.y, v.z, v.w);
   };
   uint64_t any_bit = 0;
   ulonglong2 filter_part[FILTERS][ESCAPE_SLICE_WORDS32 / 4u / ESCAPE_SLICE_THREADS];
#pragma unroll
   for (int f = 0; f < FILTERS; ++f) {  // this slice of every filter: 16 bytes per thread, zeros past the end of the row (and for a filter past the last)
      const uint32_t first_word = slice * (ESCAPE_SLICE_WORDS32 / 2u);
      const bool present = first_filter + f < n_filters;
      const uint64_t* filter = args.filters[present ? first_filter + f : first_filter];
#pragma unroll
      for (uint32_t j = 0; j < ESCAPE_SLICE_WORDS32 / 4u / ESCAPE_SLICE_THREADS; ++j) {
         const uint32_t word = first_word + (j * ESCAPE_SLICE_THREADS + threadIdx.x) * 2u;  // 16-byte chunk of the slice
         filter_part[f][j] = present && word < args.row_words ? *reinterpret_cast<const ulonglong2*>(filter + word) : make_ulonglong2(0, 0);
      }
   }
   uint4 quad0 = loadKeys(0), quad1 = loadKeys(1), quad2 = loadKeys(2);  // three granules in flight per wave, in registers of their own
   if (threadIdx.x < n_granules) {
      s_base[threadIdx.x] = range.granule_base[first_granule + threadIdx.x];
   }
   if (threadIdx.x == 64u) {  // (a key that went to the overflow list reads as the largest relative counter: a wider window, nothing else)
      s_base[n_granules] = range.granule_base[first_granule + n_granules - 1u] + (range.keys[share_end - 1u] >> ESCAPE_SLICE_SHIFT) + 1u;
   }
   // Eight filters: their slices are kept as ONE BYTE PER ROW — bit f = filter f has the row — so that a key's lookup is one
   // LDS read for all eight (a read per filter and key made the eight-filter pass LDS-bound: 32 of its ~70 LDS instructions
   // per granule and wave).  A thread holds the 128 rows of its 16-byte part of every filter and writes their 128 bytes.
   constexpr bool BYTE_PER_ROW = FILTERS == 8;
   static_assert(ESCAPE_SLICE_WORDS32 / 4u / ESCAPE_SLICE_THREADS == 1u, "a thread holds one 16-byte part of a filter slice");
#pragma unroll
   for (int f = 0; f < FILTERS; ++f) {
      for (uint32_t j = threadIdx.x * 4u; j < WINDOW; j += ESCAPE_SLICE_THREADS * 4u) {  // (16 bytes per store; WINDOW is a multiple of 4)
         *reinterpret_cast<uint4*>(s_count + f * WINDOW + j) = make_uint4(0, 0, 0, 0);
      }
      if constexpr (!BYTE_PER_ROW) {
         *reinterpret_cast<ulonglong2*>(s_filter + f * ESCAPE_SLICE_WORDS32 + threadIdx.x * 4u) = filter_part[f][0];
      }
      any_bit |= filter_part[f][0].x | filter_part[f][0].y;
   }
   if constexpr (BYTE_PER_ROW) {
#pragma unroll
      for (uint32_t quarter = 0; quarter < 4; ++quarter) {  // 32 rows of the thread's 128: 32 bytes
         uint32_t bytes[8];
#pragma unroll
         for (uint32_t k = 0; k < 8; ++k) {
            bytes[k] = 0;
         }
#pragma unroll
         for (int f = 0; f < FILTERS; ++f) {
            const uint64_t half = quarter < 2 ? filter_part[f][0].x : filter_part[f][0].y;
            const uint32_t rows32 = static_cast<uint32_t>(half >> (32u * (quarter & 1u)));
#pragma unroll
            for (uint32_t k = 0; k < 8; ++k) {  // four rows -> the low bits of four bytes
               bytes[k] |= ((((rows32 >> (4u * k)) & 0xFu) * 0x00204081u) & 0x01010101u) << f;
            }
         }
         uint32_t* out = s_filter + threadIdx.x * 32u + quarter * 8u;  // (row r of the slice = byte r)
         *reinterpret_cast<uint4*>(out) = make_uint4(bytes[0], bytes[1], bytes[2], bytes[3]);
         *reinterpret_cast<uint4*>(out + 4) = make_uint4(bytes[4], bytes[5], bytes[6], bytes[7]);
      }
   }
   if (__syncthreads_or(any_bit != 0 ? 1 : 0) == 0) {
      return;  // no row of this slice is selected: none of its keys counts
   }
   const uint32_t lane = __lane_id();
   // the chunk being counted: the granules up to chunk_granules, its window of counters
   uint32_t window_first = 0, window_used = 0, chunk_granules = 0, chunk_end = 0;
   const auto beginChunk = [&](uint32_t g) {
      // the window begins at the chunk's first key's position (the range's first position where the granule begins before it)
      // and takes the granules that end within WINDOW counters of that, one at least
      const uint32_t first_counter = max(s_base[g], range_first);
      window_first = first_counter / args.out_symbols * args.out_symbols - range_first;
      uint32_t h = g + 1u;
      while (h < n_granules && s_base[h + 1u] - range_first - window_first < WINDOW) {  // (a granule's last key may sit on the next one's first counter)
         ++h;
      }
      chunk_granules = h;
      chunk_end = min(share_begin + h * ESCAPE_GRANULE_KEYS, share_end);
      window_used = min(WINDOW, (s_base[h] / args.out_symbols + 1u) * args.out_symbols - range_first - window_first);
   };
   // the chunk's window goes to the table — contiguous atomics, 64 consecutive counters per wave instruction — and is zero
   // again for the next chunk
   const auto flushChunk = [&]() {
      ldsBarrier();
#pragma unroll
      for (int f = 0; f < FILTERS; ++f) {
         uint32_t* __restrict__ counts = range.counts[first_filter + f < n_filters ? first_filter + f : first_filter] + window_first;
         for (uint32_t j = threadIdx.x; j < window_used; j += ESCAPE_SLICE_THREADS) {
            const uint32_t value = s_count[f * WINDOW + j];
            if (value != 0) {
               s_count[f * WINDOW + j] = 0;
               atomicAdd(&counts[j], value);
            }
         }
      }
      ldsBarrier();
   };
   const auto countGranule = [&](uint4& in_flight, uint32_t g) {  // (g is uniform)
      if (g >= n_granules) {
         return;
      }
      if (g == chunk_granules) {
         flushChunk();
         beginChunk(g);
      }
      const uint4 quad = in_flight;
      in_flight = loadKeys(g + 3u);
      const uint32_t granule_first = share_begin + g * ESCAPE_GRANULE_KEYS;
      {
         const uint32_t granule_counter = s_base[g] - range_first - window_first;  // (wraps below the window: such keys are masked)
         const uint32_t i = granule_first + threadIdx.x * 4u;
         const uint32_t keys4[4] = {quad.x, quad.y, quad.z, quad.w};
         uint32_t in_window[4];
         bool valid[4];
         // the keys before the scanned positions' first and behind their last, read along in the first and the last granule, are masked out
         // (one unsigned comparison per key: index - first valid index < number of valid indices)
         if (granule_first >= key_begin && granule_first + ESCAPE_GRANULE_KEYS <= chunk_end) {  // (uniform) the granule lies inside: nearly all do
#pragma unroll
            for (uint32_t c = 0; c < 4; ++c) {
               valid[c] = keys4[c] != ESCAPE_KEY_INVALID;
            }
         } else {
#pragma unroll
            for (uint32_t c = 0; c < 4; ++c) {
               valid[c] = static_cast<bool>(static_cast<uint32_t>(keys4[c] != ESCAPE_KEY_INVALID) & static_cast<uint32_t>(i + c - key_begin < chunk_end - key_begin));
            }
         }
#pragma unroll
         for (uint32_t c = 0; c < 4; ++c) {
            in_window[c] = granule_counter + (keys4[c] >> ESCAPE_SLICE_SHIFT);
         }
         // A lane's four keys are consecutive keys of the sorted list.  Those on the counter of its first key are summed in the
         // lane (n0 <= 4); across the lanes these first counters ascend, lanes on the same one form a stretch, and a stretch adds
         // ONCE, through its last lane: the selected keys of the lanes up to and including it (population counts of the wave's
         // ballots of the bits of n0) minus those before the stretch's first lane (fetched from that lane) — no 64 lanes on one
         // LDS counter (identical addresses do not combine: ~12 cycles per lane), no add at all for a stretch without a selected
         // key (the keys read along outside the chunk lie in such stretches), and ~80 instructions per four keys where a stretch
         // mask per key column took 300.  A key on another counter than the lane's first (a lane on a boundary) adds by itself.
         const uint32_t counter0 = in_window[0];
         const uint32_t previous = static_cast<uint32_t>(__builtin_amdgcn_update_dpp(0, static_cast<int>(counter0), 0x138 /* wave_shr:1 */, 0xF, 0xF, false));
         const uint32_t following = static_cast<uint32_t>(__builtin_amdgcn_update_dpp(0, static_cast<int>(counter0), 0x130 /* wave_shl:1 */, 0xF, 0xF, false));
         const bool head = lane == 0 || counter0 != previous;
         const bool tail = lane == 63 || counter0 != following;
         const uint64_t heads_at_or_below = __ballot(head) & (~uint64_t{0} >> (63u - lane));  // (lane 0 is one: never empty)
         const uint32_t first_of_stretch = 63u - static_cast<uint32_t>(__builtin_clzll(heads_at_or_below));
         // The body has no per-key branch: an add that has nothing to add goes to a word of the lane's own (64 lanes adding zero
         // to one counter would still serialise) — 4 LDS atomics per granule and filter whatever the keys.  Where a granule by
         // itself always fits the window (FILTERS <= 2: ESCAPE_MAX_RELATIVE) every selected key of a chunk lies inside it and
         // there is no second path either.  With per-key branches and a table path through a merged (flat) address the body
         // took 250 instructions per granule and wave, half of them exec-mask traffic, and the kernel was bound by them
         // (profiles/r03_notes.md): 151 now.
         constexpr bool EVERY_KEY_IN_WINDOW = WINDOW >= ESCAPE_MAX_RELATIVE + 64u;
         [[maybe_unused]] uint32_t filters_with[4] = {0, 0, 0, 0};  // (one byte per row: bit f = filter f has the key's row)
         if constexpr (BYTE_PER_ROW) {
#pragma unroll
            for (uint32_t c = 0; c < 4; ++c) {
               filters_with[c] = reinterpret_cast<const uint8_t*>(s_filter)[keys4[c] & ESCAPE_ROW_MASK] & (valid[c] ? 0xFFu : 0u);  // (the read itself is always inside the slice)
            }
         }
         if constexpr (BYTE_PER_ROW) {
            // Eight filters at once.  The lane's sums per filter (<= 4) sit two to a register in 16-bit fields, so ONE inclusive
            // scan over the lanes (6 DPP adds per register) gives every filter's prefix, and one ds_bpermute per register the
            // prefixes at the stretch's first lane; only the final adds are per filter.  (Filter by filter — ballots, mbcnt,
            // a bpermute each — the pass cost eight times the one-filter kernel per key: 2/3 of the configs[4] batch.)
            const auto add8 = [&](uint32_t counter, uint32_t value, int f) {
               const bool here = value != 0 && counter < WINDOW;
               if (__ballot(here) != 0) {
                  atomicAdd(here ? &s_count[f * WINDOW + counter] : &s_nowhere[lane], here ? value : 0u);
               }
               if (value != 0 && counter >= WINDOW) {  // a key past the window: straight to the table
                  atomicAdd(&range.counts[first_filter + f < n_filters ? first_filter + f : first_filter][window_first + counter], value);
               }
            };
            uint32_t on_first = filters_with[0];          // per key: the filters that have it, if it sits on the lane's first counter
            uint32_t sums[4] = {0, 0, 0, 0};              // [k]: filters 2k (low field) and 2k + 1 (high field)
            uint32_t elsewhere[4] = {0, 0, 0, 0};         // per key: the filters that have it, if it sits on another counter
#pragma unroll
            for (uint32_t c = 0; c < 4; ++c) {
               if (c != 0) {
                  const bool same = in_window[c] == counter0;
                  on_first = same ? filters_with[c] : 0u;
                  elsewhere[c] = same ? 0u : filters_with[c];
               }
#pragma unroll
               for (uint32_t k = 0; k < 4; ++k) {
                  sums[k] += ((on_first >> (2u * k)) & 1u) | (((on_first >> (2u * k + 1u)) & 1u) << 16);
               }
            }
#pragma unroll
            for (uint32_t k = 0; k < 4; ++k) {
               const uint32_t through = waveSumToLane63(sums[k]);  // (inclusive scan over the lanes: <= 256 per field)
               const uint32_t before = through - sums[k];
               const uint32_t before_stretch = static_cast<uint32_t>(__builtin_amdgcn_ds_bpermute(static_cast<int>(first_of_stretch * 4u), static_cast<int>(before)));
               const uint32_t stretch = through - before_stretch;  // (field by field: no borrow, a prefix never exceeds a later one)
               add8(counter0, tail ? stretch & 0xFFFFu : 0u, static_cast<int>(2u * k));
               add8(counter0, tail ? stretch >> 16 : 0u, static_cast<int>(2u * k + 1u));
            }
#pragma unroll
            for (uint32_t c = 1; c < 4; ++c) {
               if (__ballot(elsewhere[c] != 0) != 0) {  // (uniform) a lane on a boundary of counters
#pragma unroll
                  for (int f = 0; f < FILTERS; ++f) {
                     add8(in_window[c], (elsewhere[c] >> f) & 1u, f);
                  }
               }
            }
            return;
         }
#pragma unroll
         for (int f = 0; f < FILTERS; ++f) {
            uint32_t* __restrict__ window = s_count + f * WINDOW;
            // (an add of nothing goes to the lane's own word; where a key may lie past the window it goes to the table by itself)
            [[maybe_unused]] uint32_t* __restrict__ table = range.counts[first_filter + f < n_filters ? first_filter + f : first_filter] + window_first;
            const auto add = [&](uint32_t counter, uint32_t value) {
               const bool here = EVERY_KEY_IN_WINDOW ? value != 0 : value != 0 && counter < WINDOW;
               atomicAdd(here ? &window[counter] : &s_nowhere[lane], here ? value : 0u);
               if constexpr (!EVERY_KEY_IN_WINDOW) {
                  if (value != 0 && counter >= WINDOW) {
                     atomicAdd(&table[counter], value);
                  }
               }
            };
            uint32_t n0 = 0;
            uint32_t elsewhere[4] = {0, 0, 0, 0};  // a key of the lane on another counter than its first, selected
#pragma unroll
            for (uint32_t c = 0; c < 4; ++c) {
               const uint32_t row = keys4[c] & ESCAPE_ROW_MASK;
               const uint32_t selected = (s_filter[f * ESCAPE_SLICE_WORDS32 + (row >> 5)] >> (row & 31u)) & (valid[c] ? 1u : 0u);  // (the read itself is always inside the slice)
               if (c == 0) {
                  n0 = selected;
               } else {
                  const bool same = in_window[c] == counter0;
                  n0 += same ? selected : 0u;
                  elsewhere[c] = same ? 0u : selected;
               }
            }
            if (__ballot((elsewhere[1] | elsewhere[2] | elsewhere[3]) != 0) != 0) {  // (uniform: where the keys are many to a counter no lane has one)
#pragma unroll
               for (uint32_t c = 1; c < 4; ++c) {
                  add(in_window[c], elsewhere[c]);
               }
            }
            // the stretch's sum at its last lane: an inclusive scan of the lanes' sums (6 DPP adds) less the prefix at its first lane
            const uint32_t through = waveSumToLane63(n0);
            const uint32_t before_stretch = static_cast<uint32_t>(__builtin_amdgcn_ds_bpermute(static_cast<int>(first_of_stretch * 4u), static_cast<int>(through - n0)));
            add(counter0, tail ? through - before_stretch : 0u);
         }
      }
   };
   beginChunk(0);
   for (uint32_t g = 0; g < n_granules; g += 3u) {  // (uniform)
      countGranule(quad0, g);
      countGranule(quad1, g + 1u);
      countGranule(quad2, g + 2u);
   }
   flushChunk();
}

/// The few keys of a store that do not fit the packed form (SeqStoreHost::Layout::d_escapes_overflow: counter << 32 | sequence),
/// for the positions [pos_begin, pos_end): one global filter lookup and one atomic each; grid.y = filter.
__global__ __launch_bounds__(256) void k_scan_escapes_overflow(
   const uint64_t* __restrict__ keys, uint32_t n_keys, const ScanBatchArgs batch, uint32_t pos_begin, uint32_t pos_end
) {
   const uint32_t q = blockIdx.y;
   const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
   if (i >= n_keys) {
      return;
   }
   const uint64_t key = keys[i];
   const uint32_t counter = static_cast<uint32_t>(key >> 32);
   const uint32_t sequence = static_cast<uint32_t>(key);
   if (counter >= pos_begin * batch.out_symbols && counter < pos_end * batch.out_symbols && ((batch.filters[q][sequence >> 6] >> (sequence & 63u)) & 1ull) != 0) {
      atomicAdd(&batch.counts[0][q][counter - pos_begin * batch.out_symbols], 1u);
   }
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
   // k_scan_missing_runs with the diff in LDS: a block hands its diff over as a part — [filter][range][slice][block of the slice]
   // x part_stride words, plain stores — and raises its flag (zeroed by the prepare step); k_sum_run_parts adds the parts up
   uint32_t* run_parts;
   uint32_t* run_flags;
   uint32_t part_stride;
   uint32_t run_blocks_per_slice;
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
   // the slice's runs are dealt to the gridDim.x blocks of the slice in chunks of RUNS_IN_FLIGHT x 1024; a chunk's loads are
   // in flight together, and the next chunk's while this one is counted (the first beside the filter slice)
   const uint32_t chunk_runs = DERIVED_THREADS * RUNS_IN_FLIGHT;
   const auto loadRuns = [&](uint64_t (&key)[RUNS_IN_FLIGHT], uint32_t (&run_last)[RUNS_IN_FLIGHT], uint32_t base) {
#pragma unroll
      for (uint32_t k = 0; k < RUNS_IN_FLIGHT; ++k) {
         const uint32_t i = base + k * DERIVED_THREADS + threadIdx.x;
         key[k] = i < run_end ? range.run_keys[i] : 0;
         run_last[k] = i < run_end ? range.run_ends[i] : 0;  // (an empty run: start >= end below)
      }
   };
   uint64_t any_bit = 0;
   ulonglong2 filter_part[ESCAPE_SLICE_WORDS32 / 4u / DERIVED_THREADS];
   {
      const uint64_t* filter = args.filters[q];
      const uint32_t first_word = slice * (ESCAPE_SLICE_WORDS32 / 2u);
#pragma unroll
      for (uint32_t j = 0; j < ESCAPE_SLICE_WORDS32 / 4u / DERIVED_THREADS; ++j) {
         const uint32_t word = first_word + (j * DERIVED_THREADS + threadIdx.x) * 2u;  // 16-byte chunk of the slice
         filter_part[j] = word < args.row_words ? *reinterpret_cast<const ulonglong2*>(filter + word) : make_ulonglong2(0, 0);
      }
   }
   uint64_t next_key[RUNS_IN_FLIGHT];
   uint32_t next_last[RUNS_IN_FLIGHT];
   loadRuns(next_key, next_last, run_begin + blockIdx.x * chunk_runs);
   const uint32_t n = range.n_positions;
   if constexpr (LDS_DIFF) {
      for (uint32_t j = threadIdx.x * 4u; j <= n; j += DERIVED_THREADS * 4u) {  // (16 bytes per store; the array is rounded up to them)
         *reinterpret_cast<uint4*>(s_diff + j) = make_uint4(0, 0, 0, 0);
      }
   }
#pragma unroll
   for (uint32_t j = 0; j < ESCAPE_SLICE_WORDS32 / 4u / DERIVED_THREADS; ++j) {
      *reinterpret_cast<ulonglong2*>(s_runs + (j * DERIVED_THREADS + threadIdx.x) * 4u) = filter_part[j];
      any_bit |= filter_part[j].x | filter_part[j].y;
   }
   if (__syncthreads_or(any_bit != 0 ? 1 : 0) == 0) {
      return;  // no row of this slice is selected
   }
   uint32_t* __restrict__ diff = range.scratch + static_cast<size_t>(q) * range.stride + static_cast<size_t>(n) * range.n_scan;
   const uint32_t slice_first_row = slice << ESCAPE_SLICE_SHIFT;
   const uint32_t pos_end = range.pos_begin + n;
   uint32_t from_the_first = 0;  // selected runs that begin at or before the range's first position (sequences that begin with the missing symbol: every lane on one counter otherwise)
   for (uint32_t base = run_begin + blockIdx.x * chunk_runs; base < run_end; base += gridDim.x * chunk_runs) {
      uint64_t key[RUNS_IN_FLIGHT];
      uint32_t run_last[RUNS_IN_FLIGHT];
#pragma unroll
      for (uint32_t k = 0; k < RUNS_IN_FLIGHT; ++k) {
         key[k] = next_key[k];
         run_last[k] = next_last[k];
      }
      if (base + gridDim.x * chunk_runs < run_end) {  // (uniform)
         loadRuns(next_key, next_last, base + gridDim.x * chunk_runs);
      }
#pragma unroll
      for (uint32_t k = 0; k < RUNS_IN_FLIGHT; ++k) {
         const uint32_t local = (static_cast<uint32_t>(key[k] >> 32) - slice_first_row) & ((1u << ESCAPE_SLICE_SHIFT) - 1u);
         const bool selected = ((s_runs[local >> 5] >> (local & 31u)) & 1u) != 0;
         const uint32_t start = max(static_cast<uint32_t>(key[k]), range.pos_begin);
         const uint32_t end = min(run_last[k], pos_end);
         if (selected && start < end) {
            if (start == range.pos_begin) {
               from_the_first += 1;
            } else if constexpr (LDS_DIFF) {
               atomicAdd(&s_diff[start - range.pos_begin], 1u);
            } else {
               atomicAdd(&diff[start - range.pos_begin], 1u);
            }
            if (end < pos_end) {  // (the entry behind the last position is never summed)
               if constexpr (LDS_DIFF) {
                  atomicAdd(&s_diff[end - range.pos_begin], 0xFFFFFFFFu);
               } else {
                  atomicAdd(&diff[end - range.pos_begin], 0xFFFFFFFFu);
               }
            }
         }
      }
   }
   from_the_first = waveSumToLane63(from_the_first);
   if ((threadIdx.x & 63u) == 63u && from_the_first != 0) {
      if constexpr (LDS_DIFF) {
         atomicAdd(&s_diff[0], from_the_first);
      } else {
         atomicAdd(&diff[0], from_the_first);
      }
   }
   if constexpr (LDS_DIFF) {
      // The block's diff leaves as a part of its own, in plain 16-byte stores; k_sum_run_parts adds the parts up.  (Adding it
      // to the range's diff with atomics from here — 231 blocks x 30 000 entries at 10 M rows, device-scope atomics are
      // performed at the memory side — took 30 of this kernel's 43 us: profiles/r03_notes.md.)
      __syncthreads();
      const uint32_t part = ((q * args.n_ranges + blockIdx.y / args.n_run_slices) * args.n_run_slices + slice) * gridDim.x + blockIdx.x;
      uint32_t* __restrict__ out = args.run_parts + static_cast<size_t>(part) * args.part_stride;
      for (uint32_t j = threadIdx.x * 4u; j <= n; j += DERIVED_THREADS * 4u) {
         *reinterpret_cast<uint4*>(out + j) = *reinterpret_cast<const uint4*>(s_diff + j);
      }
      if (threadIdx.x == 0) {
         args.run_flags[part] = 1u;
      }
   }
}

/// diff[j] of a range and filter += the parts of the blocks of k_scan_missing_runs that raised their flag.  grid = (blocks of
/// 1024 entries, range x RUN_PART_GROUPS, filter): a thread owns 4 consecutive entries and a group of parts.
constexpr uint32_t RUN_PART_GROUPS = 16;
__global__ __launch_bounds__(256) void k_sum_run_parts(const DerivedArgs args) {
   const uint32_t q = blockIdx.z;
   const uint32_t r = blockIdx.y / RUN_PART_GROUPS;
   const uint32_t group = blockIdx.y % RUN_PART_GROUPS;
   const DerivedRange& range = args.ranges[r];
   const uint32_t n = range.n_positions;
   const uint32_t j = (blockIdx.x * 256u + threadIdx.x) * 4u;
   if (range.code_map == nullptr || blockIdx.x * 1024u > n) {
      return;  // (uniform)
   }
   const uint32_t parts_of_range = args.n_run_slices * args.run_blocks_per_slice;
   const uint32_t per_group = (parts_of_range + RUN_PART_GROUPS - 1) / RUN_PART_GROUPS;
   const uint32_t first = (q * args.n_ranges + r) * parts_of_range;
   const uint32_t begin = first + group * per_group;
   const uint32_t end = min(begin + per_group, first + parts_of_range);
   const uint32_t j_safe = j <= n ? j : 0;
   uint4 sum = make_uint4(0, 0, 0, 0);
   for (uint32_t part = begin; part < end; part += 8) {  // (uniform) eight parts' loads in flight
      uint4 v[8];
#pragma unroll
      for (uint32_t k = 0; k < 8; ++k) {
         v[k] = make_uint4(0, 0, 0, 0);
         if (part + k < end && args.run_flags[part + k] != 0) {
            v[k] = *reinterpret_cast<const uint4*>(args.run_parts + static_cast<size_t>(part + k) * args.part_stride + j_safe);
         }
      }
#pragma unroll
      for (uint32_t k = 0; k < 8; ++k) {
         sum.x += v[k].x;
         sum.y += v[k].y;
         sum.z += v[k].z;
         sum.w += v[k].w;
      }
   }
   if (j > n) {
      return;
   }
   uint32_t* __restrict__ diff = range.scratch + static_cast<size_t>(q) * range.stride + static_cast<size_t>(n) * range.n_scan;
   const uint32_t values[4] = {sum.x, sum.y, sum.z, sum.w};
#pragma unroll
   for (uint32_t c = 0; c < 4; ++c) {
      if (values[c] != 0 && j + c <= n) {
         atomicAdd(&diff[j + c], values[c]);
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
   uint64_t word[SPARSE_KEYS_PER_THREAD];
#pragma unroll
   for (uint32_t k = 0; k < SPARSE_KEYS_PER_THREAD; ++k) {  // the filter lookups of all keys of the thread in flight together
      word[k] = args.filters[q][static_cast<uint32_t>(key[k]) >> 6];
   }
#pragma unroll
   for (uint32_t k = 0; k < SPARSE_KEYS_PER_THREAD; ++k) {
      const uint32_t sequence = static_cast<uint32_t>(key[k]);
      bool pending = first + k * 256u < range.sparse_end && ((word[k] >> (sequence & 63u)) & 1ull) != 0;
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

/// A position range of one sequence store with the count tables of every filter of the launch.
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
         (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_scan_escapes_sliced<2>), hipFuncAttributeMaxDynamicSharedMemorySize, escapeLdsBytes<2>());
         (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_scan_escapes_sliced<4>), hipFuncAttributeMaxDynamicSharedMemorySize, escapeLdsBytes<4>());
         (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_scan_escapes_sliced<8>), hipFuncAttributeMaxDynamicSharedMemorySize, escapeLdsBytes<8>());
      });
      const uint32_t per_block = q_count <= 1 ? 1 : (q_count <= 2 ? 2 : (q_count <= 4 ? 4 : 8));  // filters per pass over the keys
      // a block's share of a slice's keys: whole granules, about five shares to each of the 512 places of the chip (the slices
      // differ in their keys, and a share's time in how its keys lie), at most ESCAPE_GRANULES_PER_BLOCK
      const uint32_t passes = (q_count + per_block - 1) / per_block;
      const uint64_t granules = (total_keys + ESCAPE_GRANULE_KEYS - 1) / ESCAPE_GRANULE_KEYS * passes;
      const uint32_t block_granules = static_cast<uint32_t>(std::min<uint64_t>(ESCAPE_GRANULES_PER_BLOCK, std::max<uint64_t>(1, granules / 1280)));  // (flat between 640 and 2 560: profiles/r03_notes.md)
      const uint32_t block_keys = block_granules * ESCAPE_GRANULE_KEYS;
      sliced.block_keys = block_keys;
      const dim3 grid((most_keys + block_keys - 1) / block_keys, sliced.n_slices * n_sliced, passes);
      char name[64];
      std::snprintf(name, sizeof(name), "k_scan_escapes_sliced<%u>", per_block);
      // bytes: the keys (4 each) once per pass of `per_block` filters, plus a 16 KiB filter slice per block and filter
      ScanLaunchTiming* timing = startLaunchTiming(
         name, 0, total_keys * sizeof(uint32_t) * grid.z + static_cast<uint64_t>(grid.x) * grid.y * q_count * ESCAPE_SLICE_WORDS32 * sizeof(uint32_t), q_count,
         grid.x * grid.y * grid.z, hip_stream
      );
      switch (per_block) {
         case 1: k_scan_escapes_sliced<1><<<grid, ESCAPE_SLICE_THREADS, escapeLdsBytes<1>(), hip_stream>>>(sliced, q_count); break;
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
         entry.granule_base = layout.d_granule_base;
         entry.slice_first = layout.d_slice_first;
         if (layout.n_overflow != 0) {  // the few keys that do not fit the packed form: a small launch of their own
            ScanBatchArgs overflow{};
            overflow.out_symbols = range.seqstore->dev.n_scan;
            for (uint32_t q = 0; q < q_count; ++q) {
               overflow.filters[q] = filters[q];
               overflow.counts[0][q] = range.counts[q];
            }
            k_scan_escapes_overflow<<<dim3((layout.n_overflow + 255) / 256, q_count), 256, 0, hip_stream>>>(
               layout.d_escapes_overflow, layout.n_overflow, overflow, range.pos_begin, range.pos_end
            );
            HIP_TRY(hipGetLastError());
         }
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
            most_keys = std::max(most_keys, keys + ESCAPE_GRANULE_KEYS - 1u);  // (blocks start at a granule boundary)
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
   size_t table_words = 0;       // zeroed by the prepare step: the tables, then the flags of the run parts
   size_t part_words = 0;        // behind them, not zeroed: the run parts (k_scan_missing_runs -> k_sum_run_parts)
   uint32_t most_positions = 0;  // of a range with derived symbols
};

/// Blocks per slice of k_scan_missing_runs: one block per CU fits (its LDS), about one round of the 256 CUs over all (slice, range, filter).
uint32_t runBlocksPerSlice(const DerivedArgs& launch, uint32_t q_count) {
   const uint32_t run_units = std::max<uint32_t>(1, launch.n_run_slices * launch.n_ranges * q_count);
   return std::min<uint32_t>(8, std::max<uint32_t>(1, 240 / run_units));
}

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
   // the parts of the blocks of k_scan_missing_runs: flags in the zeroed area, the parts behind it (offsets until bindDerived)
   const uint32_t part_stride = (plan.most_positions + 4) / 4 * 4;
   size_t part_offset = 0;
   for (DerivedArgs& launch : plan.launches) {
      launch.run_blocks_per_slice = runBlocksPerSlice(launch, q_count);
      launch.part_stride = part_stride;
      const size_t parts = static_cast<size_t>(q_count) * launch.n_ranges * launch.n_run_slices * launch.run_blocks_per_slice;
      launch.run_flags = reinterpret_cast<uint32_t*>(offset * sizeof(uint32_t));
      offset += (parts + 3) / 4 * 4;
      launch.run_parts = reinterpret_cast<uint32_t*>(part_offset * sizeof(uint32_t));
      part_offset += parts * part_stride;
   }
   plan.table_words = offset;
   plan.part_words = part_offset;
}

/// The tables get their place in the scratch block; the private ranges point at them.
void bindDerived(DerivedPlan& plan, const SparseScratch& scratch, uint32_t q_count) {
   size_t r = 0;
   for (DerivedArgs& launch : plan.launches) {
      launch.counters = scratch.counters[scratch.set];
      launch.run_flags = scratch.tables + reinterpret_cast<size_t>(launch.run_flags) / sizeof(uint32_t);
      launch.run_parts = scratch.tables + plan.table_words + reinterpret_cast<size_t>(launch.run_parts) / sizeof(uint32_t);
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
      const size_t lds_bytes = (ESCAPE_SLICE_WORDS32 + (static_cast<size_t>(plan.most_positions) + 4) / 4 * 4) * sizeof(uint32_t);
      const bool lds_diff = lds_bytes <= 152 * 1024;
      static std::once_flag lds_once;
      std::call_once(lds_once, [] {
         (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_scan_missing_runs<true>), hipFuncAttributeMaxDynamicSharedMemorySize, 152 * 1024);
      });
      const dim3 run_grid(launch.run_blocks_per_slice, launch.n_run_slices * launch.n_ranges, q_count);
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
         k_sum_run_parts<<<dim3(plan.most_positions / 1024 + 1, launch.n_ranges * RUN_PART_GROUPS, q_count), 256, 0, hip_stream>>>(launch);
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
int scanRangesImpl(
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
   const int acquired = acquireSparseScratch(store->device, capacity, plan.table_words + plan.part_words, &scratch);
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


}  // namespace

namespace silo_gpu_detail {
int scanRanges(const silo_gpu_store* store, const std::vector<ScanRange>& ranges, const uint64_t* const* filters, uint32_t q_count, hipStream_t hip_stream) {
   return scanRangesImpl(store, ranges, filters, q_count, hip_stream);
}
}  // namespace silo_gpu_detail

extern "C" {

const char* silo_gpu_last_scan_kernel(void) {
   return g_last_scan_kernel;
}

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


}  // extern "C"
