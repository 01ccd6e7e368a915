// silo_gpu_columns.hip — metadata columns on the device.
//
//   K5  k_bitset_from_compare   a column predicate (CompareToValueSelection<T>::match, selection.cpp:145-165) for
//                               all rows at once: bit i = values[i] <op> value; one wave ballot per bitset word
//   K6  k_group_count           Aggregated with groupByFields (aggregated.cpp:100-149): histogram of the combined
//                               dictionary ids of the filtered rows
//
// Both stream a 4- or 8-byte column once (HBM-bound, 4..8 B per row) and write 1 bit / a few counters per row.
#include <algorithm>

#include "internal.h"

namespace {

template <typename T>
__device__ __forceinline__ bool compareValues(T row_value, int comparator, T value) {
   switch (comparator) {  // plain C++ comparisons: for doubles IEEE semantics, every comparison with NaN but != is false
      case SILO_GPU_CMP_EQUALS: return row_value == value;
      case SILO_GPU_CMP_NOT_EQUALS: return row_value != value;
      case SILO_GPU_CMP_LESS: return row_value < value;
      case SILO_GPU_CMP_HIGHER_OR_EQUALS: return row_value >= value;
      case SILO_GPU_CMP_HIGHER: return row_value > value;
      default: return row_value <= value;
   }
}

// One wave per COMPARE_WORDS bitset words: lane l tests rows 64 * word + l of each of them — COMPARE_WORDS
// independent, fully coalesced 256- or 512-byte reads in flight per wave — and the ballot of each test is the
// word.  Words past the last row (row padding) are written as zero.
constexpr uint32_t COMPARE_WORDS = 8;

template <typename T>
__global__ __launch_bounds__(256) void k_bitset_from_compare(
   const T* __restrict__ values, uint32_t n_rows, uint32_t row_words, int comparator, T value, uint64_t* __restrict__ out
) {
   const uint32_t lane = threadIdx.x & 63u;
   const uint32_t first_word = (blockIdx.x * (blockDim.x / 64) + (threadIdx.x >> 6)) * COMPARE_WORDS;
   if (first_word >= row_words) {
      return;
   }
   T row_values[COMPARE_WORDS];
#pragma unroll
   for (uint32_t k = 0; k < COMPARE_WORDS; ++k) {
      const uint32_t row = (first_word + k) * 64 + lane;
      row_values[k] = row < n_rows ? values[row] : value;
   }
#pragma unroll
   for (uint32_t k = 0; k < COMPARE_WORDS; ++k) {
      const uint32_t row = (first_word + k) * 64 + lane;
      const bool match = row < n_rows && compareValues<T>(row_values[k], comparator, value);
      const uint64_t ballot = __ballot(match);
      if (lane == 0 && first_word + k < row_words) {
         out[first_word + k] = ballot;
      }
   }
}

constexpr uint32_t GROUP_LDS_BINS = 4096;
constexpr uint32_t GROUP_ROWS_PER_THREAD = 16;

struct GroupCountArgs {
   const uint32_t* ids[SILO_GPU_MAX_GROUP_COLUMNS];
   uint32_t strides[SILO_GPU_MAX_GROUP_COLUMNS];
   uint32_t n_columns;
};

// Adds one to bins[key] for every active lane.  The lanes that share the key of the first active lane are counted
// by ONE atomic (takes the sting out of skewed columns, where most rows of a wave fall into one group); the rest
// add individually.
template <typename Add>
__device__ __forceinline__ void aggregatedIncrement(bool active, uint32_t key, Add add) {
   const uint64_t pending = __ballot(active);
   if (pending == 0) {
      return;
   }
   const int leader = __ffsll(static_cast<long long>(pending)) - 1;
   const uint32_t leader_key = __shfl(key, leader);
   const bool same = active && key == leader_key;
   const uint64_t group = __ballot(same);
   if (static_cast<int>(threadIdx.x & 63u) == leader) {
      add(leader_key, static_cast<uint32_t>(__popcll(group)));
   } else if (active && !same) {
      add(key, 1u);
   }
}

// Blocks walk the rows in chunks of 256 * GROUP_ROWS_PER_THREAD (grid-stride, about two blocks per CU, so the
// per-block flush of the LDS table is paid a few hundred times, not once per chunk); the ids of all of a thread's
// rows of a chunk are fetched first (GROUP_ROWS_PER_THREAD independent coalesced loads per column in flight), then
// counted.
//   DENSE  (n_bins <= GROUP_LDS_BINS): the whole histogram lives in LDS, in `copies` interleaved replicas
//          (bin * copies + lane % copies) so that lanes of a wave that hit the same small bin use different banks.
//   !DENSE (up to SILO_GPU_MAX_GROUP_BINS bins): LDS holds a direct-mapped cache key -> count; a key that loses
//          its slot to another key goes to the global table at once.  Hot groups stay in LDS and reach HBM once
//          per block.
template <bool DENSE>
__global__ __launch_bounds__(256) void k_group_count(
   const uint64_t* __restrict__ filter, uint32_t n_rows, const GroupCountArgs args, uint32_t n_bins, uint32_t copies,
   uint32_t* __restrict__ counts
) {
   __shared__ uint32_t s_counts[GROUP_LDS_BINS];
   __shared__ uint32_t s_keys[DENSE ? 1 : GROUP_LDS_BINS];  // key + 1, 0 = free
   for (uint32_t slot = threadIdx.x; slot < GROUP_LDS_BINS; slot += blockDim.x) {
      s_counts[slot] = 0;
      if (!DENSE) {
         s_keys[slot] = 0;
      }
   }
   __syncthreads();
   const uint32_t lane_copy = (threadIdx.x & 63u) & (copies - 1);
   const auto add = [&](uint32_t key, uint32_t n) {
      if (DENSE) {
         atomicAdd(&s_counts[key * copies + lane_copy], n);
      } else {
         const uint32_t slot = (key * 2654435761u) >> 20;  // 12 bits: GROUP_LDS_BINS slots
         const uint32_t owner = atomicCAS(&s_keys[slot], 0u, key + 1);
         if (owner == 0 || owner == key + 1) {
            atomicAdd(&s_counts[slot], n);
         } else {
            atomicAdd(&counts[key], n);
         }
      }
   };
   constexpr uint32_t CHUNK = 256 * GROUP_ROWS_PER_THREAD;
   for (uint32_t chunk_first = blockIdx.x * CHUNK; chunk_first < n_rows; chunk_first += gridDim.x * CHUNK) {
      uint32_t keys[GROUP_ROWS_PER_THREAD];
      uint32_t active_mask = 0;
#pragma unroll
      for (uint32_t step = 0; step < GROUP_ROWS_PER_THREAD; ++step) {
         const uint32_t row = chunk_first + step * blockDim.x + threadIdx.x;  // consecutive lanes, consecutive rows
         bool active = row < n_rows;
         if (active && filter != nullptr) {
            active = (filter[row >> 6] >> (row & 63u)) & 1u;
         }
         active_mask |= (active ? 1u : 0u) << step;
         keys[step] = 0;
      }
      for (uint32_t column = 0; column < args.n_columns; ++column) {
         const uint32_t* ids = args.ids[column];
         const uint32_t stride = args.strides[column];
#pragma unroll
         for (uint32_t step = 0; step < GROUP_ROWS_PER_THREAD; ++step) {
            const uint32_t row = min(chunk_first + step * blockDim.x + threadIdx.x, n_rows - 1);
            keys[step] += ids[row] * stride;
         }
      }
#pragma unroll
      for (uint32_t step = 0; step < GROUP_ROWS_PER_THREAD; ++step) {
         aggregatedIncrement((active_mask >> step) & 1u, keys[step], add);
      }
   }
   __syncthreads();
   if (DENSE) {
      for (uint32_t bin = threadIdx.x; bin < n_bins; bin += blockDim.x) {
         uint32_t n = 0;
         for (uint32_t copy = 0; copy < copies; ++copy) {
            n += s_counts[bin * copies + copy];
         }
         if (n != 0) {
            atomicAdd(&counts[bin], n);
         }
      }
   } else {
      for (uint32_t slot = threadIdx.x; slot < GROUP_LDS_BINS; slot += blockDim.x) {
         if (s_keys[slot] != 0) {
            atomicAdd(&counts[s_keys[slot] - 1], s_counts[slot]);
         }
      }
   }
}

// K6b: group-by for tuple spaces beyond SILO_GPU_MAX_GROUP_BINS: an open-addressing hash table in HBM keyed by the
// 64-bit mixed-radix tuple id (linear probing, atomicCAS on the key, atomicAdd on the count), then a compaction of
// the occupied slots.  The table has at least twice as many slots as rows, so probing terminates.
struct GroupHashArgs {
   const uint32_t* ids[SILO_GPU_MAX_GROUP_COLUMNS];
   unsigned long long strides[SILO_GPU_MAX_GROUP_COLUMNS];
   uint32_t n_columns;
};

constexpr unsigned long long GROUP_HASH_EMPTY = ~0ull;

__global__ __launch_bounds__(256) void k_group_hash_insert(
   const uint64_t* __restrict__ filter, uint32_t n_rows, const GroupHashArgs args, unsigned long long* __restrict__ table_keys,
   uint32_t* __restrict__ table_counts, uint32_t capacity_mask, uint32_t* __restrict__ overflow
) {
   const uint32_t row = blockIdx.x * blockDim.x + threadIdx.x;
   if (row >= n_rows) {
      return;
   }
   if (filter != nullptr && ((filter[row >> 6] >> (row & 63u)) & 1u) == 0) {
      return;
   }
   unsigned long long key = 0;
   for (uint32_t column = 0; column < args.n_columns; ++column) {
      key += static_cast<unsigned long long>(args.ids[column][row]) * args.strides[column];
   }
   unsigned long long mixed = key * 0x9E3779B97F4A7C15ull;
   mixed ^= mixed >> 29;
   uint32_t slot = static_cast<uint32_t>(mixed) & capacity_mask;
   // bounded probing: a table that is too small for the rows it gets (max_rows understated) is reported, never a hang
   for (uint32_t probe = 0; probe <= capacity_mask; ++probe) {
      const unsigned long long owner = atomicCAS(table_keys + slot, GROUP_HASH_EMPTY, key);
      if (owner == GROUP_HASH_EMPTY || owner == key) {
         atomicAdd(table_counts + slot, 1u);
         return;
      }
      slot = (slot + 1) & capacity_mask;
   }
   atomicExch(overflow, 1u);
}

__global__ __launch_bounds__(256) void k_group_hash_compact(
   const unsigned long long* __restrict__ table_keys, const uint32_t* __restrict__ table_counts, uint32_t capacity,
   unsigned long long* __restrict__ out_keys, uint32_t* __restrict__ out_counts, uint32_t max_out, uint32_t* __restrict__ n_out
) {
   const uint32_t slot = blockIdx.x * blockDim.x + threadIdx.x;
   if (slot < capacity && table_keys[slot] != GROUP_HASH_EMPTY) {
      const uint32_t index = atomicAdd(n_out, 1u);
      if (index < max_out) {  // more groups than rows promised by the caller: counted, not stored
         out_keys[index] = table_keys[slot];
         out_counts[index] = table_counts[slot];
      }
   }
}

// Insertion index (insertion_index.cpp): the occurrences of a column's insertions as pairs (row, distinct
// insertion id).  K8 marks the rows of the pairs whose insertion matched the search pattern (the regex runs on the
// host over the distinct insertions of one position, insertion_index.cpp:128-137); K9 counts, per distinct insertion,
// the pairs whose row is in the filter (insertions.cpp:196-206 and_cardinality per insertion).
__global__ __launch_bounds__(256) void k_bitset_from_pairs(
   const uint32_t* __restrict__ rows, const uint32_t* __restrict__ ids, uint32_t n_pairs, const uint8_t* __restrict__ membership,
   uint64_t* __restrict__ out
) {
   const uint32_t pair = blockIdx.x * blockDim.x + threadIdx.x;
   if (pair < n_pairs && membership[ids[pair]] != 0) {
      const uint32_t row = rows[pair];
      atomicOr(reinterpret_cast<unsigned long long*>(out + (row >> 6)), 1ull << (row & 63u));
   }
}

__global__ __launch_bounds__(256) void k_count_pairs(
   const uint64_t* __restrict__ filter, const uint32_t* __restrict__ rows, const uint32_t* __restrict__ ids, uint32_t n_pairs,
   uint32_t* __restrict__ counts
) {
   const uint32_t pair = blockIdx.x * blockDim.x + threadIdx.x;
   if (pair < n_pairs) {
      const uint32_t row = rows[pair];
      if (filter == nullptr || ((filter[row >> 6] >> (row & 63u)) & 1u) != 0) {
         atomicAdd(&counts[ids[pair]], 1u);
      }
   }
}

}  // namespace

extern "C" {

int silo_gpu_upload_column(const void* src_host, size_t n_rows, int value_type, void** out_dev) {
   if (src_host == nullptr || out_dev == nullptr || n_rows == 0) {
      return silo_gpu_internal_fail(SILO_GPU_ERR_INVALID_ARGUMENT, "silo_gpu_upload_column: bad arguments");
   }
   const size_t width = value_type == SILO_GPU_VALUE_F64 ? 8 : 4;
   return silo_gpu_upload_bytes(src_host, n_rows * width, out_dev);
}

int silo_gpu_bitset_from_compare(
   const silo_gpu_store* store, uint64_t* dst_dev, const void* values_dev, int value_type, int comparator, const void* value, void* stream
) {
   if (store == nullptr || dst_dev == nullptr || values_dev == nullptr || value == nullptr || comparator < SILO_GPU_CMP_EQUALS ||
       comparator > SILO_GPU_CMP_LESS_OR_EQUALS) {
      return silo_gpu_internal_fail(SILO_GPU_ERR_INVALID_ARGUMENT, "silo_gpu_bitset_from_compare: bad arguments");
   }
   const uint32_t n_rows = silo_gpu_store_sequence_count(store);
   const uint32_t row_words = silo_gpu_store_row_words(store);
   auto hip_stream = static_cast<hipStream_t>(stream);
   const dim3 grid((row_words + 4 * COMPARE_WORDS - 1) / (4 * COMPARE_WORDS));
   switch (value_type) {
      case SILO_GPU_VALUE_I32:
         k_bitset_from_compare<int32_t><<<grid, 256, 0, hip_stream>>>(
            static_cast<const int32_t*>(values_dev), n_rows, row_words, comparator, *static_cast<const int32_t*>(value), dst_dev
         );
         break;
      case SILO_GPU_VALUE_U32:
         k_bitset_from_compare<uint32_t><<<grid, 256, 0, hip_stream>>>(
            static_cast<const uint32_t*>(values_dev), n_rows, row_words, comparator, *static_cast<const uint32_t*>(value), dst_dev
         );
         break;
      case SILO_GPU_VALUE_F64:
         k_bitset_from_compare<double><<<grid, 256, 0, hip_stream>>>(
            static_cast<const double*>(values_dev), n_rows, row_words, comparator, *static_cast<const double*>(value), dst_dev
         );
         break;
      default: return silo_gpu_internal_fail(SILO_GPU_ERR_INVALID_ARGUMENT, "silo_gpu_bitset_from_compare: unknown value type");
   }
   SILO_HIP_TRY(hipGetLastError());
   return SILO_GPU_OK;
}

int silo_gpu_group_count(
   const silo_gpu_store* store, const uint64_t* filter_dev, const uint32_t* const* group_ids_dev, const uint32_t* cardinalities,
   uint32_t n_columns, uint32_t* counts_dev, void* stream
) {
   if (store == nullptr || group_ids_dev == nullptr || cardinalities == nullptr || counts_dev == nullptr || n_columns == 0 ||
       n_columns > SILO_GPU_MAX_GROUP_COLUMNS) {
      return silo_gpu_internal_fail(SILO_GPU_ERR_INVALID_ARGUMENT, "silo_gpu_group_count: bad arguments");
   }
   GroupCountArgs args{};
   args.n_columns = n_columns;
   uint64_t n_bins = 1;
   for (uint32_t column = 0; column < n_columns; ++column) {  // mixed radix, first column most significant
      if (group_ids_dev[column] == nullptr || cardinalities[column] == 0) {
         return silo_gpu_internal_fail(SILO_GPU_ERR_INVALID_ARGUMENT, "silo_gpu_group_count: null column or empty dictionary");
      }
      n_bins *= cardinalities[column];
      if (n_bins > SILO_GPU_MAX_GROUP_BINS) {
         return silo_gpu_internal_fail(SILO_GPU_ERR_INVALID_ARGUMENT, "silo_gpu_group_count: more than SILO_GPU_MAX_GROUP_BINS groups");
      }
   }
   uint32_t stride = 1;
   for (uint32_t column = n_columns; column-- > 0;) {
      args.ids[column] = group_ids_dev[column];
      args.strides[column] = stride;
      stride *= cardinalities[column];
   }
   const uint32_t n_rows = silo_gpu_store_sequence_count(store);
   if (n_rows == 0) {
      return SILO_GPU_OK;
   }
   auto hip_stream = static_cast<hipStream_t>(stream);
   const uint32_t rows_per_chunk = 256 * GROUP_ROWS_PER_THREAD;
   const uint32_t n_chunks = (n_rows + rows_per_chunk - 1) / rows_per_chunk;
   const dim3 grid(std::min<uint32_t>(n_chunks, 512));  // ~2 blocks per CU, each walks its share of the chunks
   if (n_bins <= GROUP_LDS_BINS) {
      uint32_t copies = 1;  // replicas of the LDS histogram: the largest power of two <= min(32, GROUP_LDS_BINS / n_bins)
      while (copies < 32 && static_cast<uint64_t>(copies) * 2 * n_bins <= GROUP_LDS_BINS) {
         copies *= 2;
      }
      k_group_count<true><<<grid, 256, 0, hip_stream>>>(filter_dev, n_rows, args, static_cast<uint32_t>(n_bins), copies, counts_dev);
   } else {
      k_group_count<false><<<grid, 256, 0, hip_stream>>>(filter_dev, n_rows, args, static_cast<uint32_t>(n_bins), 1, counts_dev);
   }
   SILO_HIP_TRY(hipGetLastError());
   return SILO_GPU_OK;
}

int silo_gpu_group_count_hashed(
   const silo_gpu_store* store, const uint64_t* filter_dev, const uint32_t* const* group_ids_dev, const uint32_t* cardinalities,
   uint32_t n_columns, uint32_t max_rows, uint64_t** out_keys_dev, uint32_t** out_counts_dev, uint32_t* out_n_groups, void* stream
) {
   if (store == nullptr || group_ids_dev == nullptr || cardinalities == nullptr || out_keys_dev == nullptr || out_counts_dev == nullptr ||
       out_n_groups == nullptr || n_columns == 0 || n_columns > SILO_GPU_MAX_GROUP_COLUMNS) {
      return silo_gpu_internal_fail(SILO_GPU_ERR_INVALID_ARGUMENT, "silo_gpu_group_count_hashed: bad arguments");
   }
   *out_keys_dev = nullptr;
   *out_counts_dev = nullptr;
   *out_n_groups = 0;
   GroupHashArgs args{};
   args.n_columns = n_columns;
   unsigned long long stride = 1;
   for (uint32_t column = n_columns; column-- > 0;) {  // mixed radix, first column most significant
      if (group_ids_dev[column] == nullptr || cardinalities[column] == 0) {
         return silo_gpu_internal_fail(SILO_GPU_ERR_INVALID_ARGUMENT, "silo_gpu_group_count_hashed: null column or empty dictionary");
      }
      args.ids[column] = group_ids_dev[column];
      args.strides[column] = stride;
      if (stride > (~0ull - 1) / cardinalities[column]) {
         return silo_gpu_internal_fail(SILO_GPU_ERR_INVALID_ARGUMENT, "silo_gpu_group_count_hashed: the tuple space exceeds 64 bits");
      }
      stride *= cardinalities[column];
   }
   const uint32_t n_rows = silo_gpu_store_sequence_count(store);
   const uint32_t rows_bound = std::min(max_rows, n_rows);
   if (rows_bound == 0) {
      return SILO_GPU_OK;
   }
   uint64_t capacity = 1024;
   while (capacity < 2ull * rows_bound) {
      capacity *= 2;
   }
   auto hip_stream = static_cast<hipStream_t>(stream);
   unsigned long long* table_keys = nullptr;
   uint32_t* table_counts = nullptr;
   unsigned long long* keys = nullptr;
   uint32_t* counts = nullptr;
   uint32_t* n_out = nullptr;
   const auto release = [&](bool keep_outputs) {
      (void)hipFree(table_keys);
      (void)hipFree(table_counts);
      (void)hipFree(n_out);
      if (!keep_outputs) {
         (void)hipFree(keys);
         (void)hipFree(counts);
      }
   };
   hipError_t err = hipMalloc(&table_keys, capacity * sizeof(unsigned long long));
   if (err == hipSuccess) err = hipMalloc(&table_counts, capacity * sizeof(uint32_t));
   if (err == hipSuccess) err = hipMalloc(&keys, static_cast<size_t>(rows_bound) * sizeof(unsigned long long));
   if (err == hipSuccess) err = hipMalloc(&counts, static_cast<size_t>(rows_bound) * sizeof(uint32_t));
   if (err == hipSuccess) err = hipMalloc(&n_out, 2 * sizeof(uint32_t));  // [0] groups, [1] overflow flag
   if (err == hipSuccess) err = hipMemsetAsync(table_keys, 0xFF, capacity * sizeof(unsigned long long), hip_stream);
   if (err == hipSuccess) err = hipMemsetAsync(table_counts, 0, capacity * sizeof(uint32_t), hip_stream);
   if (err == hipSuccess) err = hipMemsetAsync(n_out, 0, 2 * sizeof(uint32_t), hip_stream);
   if (err == hipSuccess) {
      k_group_hash_insert<<<(n_rows + 255) / 256, 256, 0, hip_stream>>>(
         filter_dev, n_rows, args, table_keys, table_counts, static_cast<uint32_t>(capacity - 1), n_out + 1
      );
      k_group_hash_compact<<<static_cast<uint32_t>((capacity + 255) / 256), 256, 0, hip_stream>>>(
         table_keys, table_counts, static_cast<uint32_t>(capacity), keys, counts, rows_bound, n_out
      );
      err = hipGetLastError();
   }
   uint32_t result[2] = {0, 0};
   if (err == hipSuccess) err = hipMemcpyAsync(result, n_out, sizeof(result), hipMemcpyDeviceToHost, hip_stream);
   if (err == hipSuccess) err = hipStreamSynchronize(hip_stream);
   if (err != hipSuccess) {
      release(false);
      SILO_HIP_TRY(err);
   }
   const uint32_t n_groups = result[0];
   if (result[1] != 0 || n_groups > rows_bound) {
      release(false);
      return silo_gpu_internal_fail(SILO_GPU_ERR_INVALID_ARGUMENT, "silo_gpu_group_count_hashed: more selected rows than max_rows");
   }
   release(true);
   *out_keys_dev = reinterpret_cast<uint64_t*>(keys);
   *out_counts_dev = counts;
   *out_n_groups = n_groups;
   return SILO_GPU_OK;
}

int silo_gpu_bitset_from_pairs(
   const silo_gpu_store* store, uint64_t* dst_dev, const uint32_t* rows_dev, const uint32_t* ids_dev, uint32_t n_pairs,
   const uint8_t* membership_by_id, uint32_t n_ids, void* stream
) {
   if (store == nullptr || dst_dev == nullptr || (n_pairs != 0 && (rows_dev == nullptr || ids_dev == nullptr || membership_by_id == nullptr || n_ids == 0))) {
      return silo_gpu_internal_fail(SILO_GPU_ERR_INVALID_ARGUMENT, "silo_gpu_bitset_from_pairs: bad arguments");
   }
   auto hip_stream = static_cast<hipStream_t>(stream);
   SILO_HIP_TRY(hipMemsetAsync(dst_dev, 0, static_cast<size_t>(silo_gpu_store_row_words(store)) * sizeof(uint64_t), hip_stream));
   if (n_pairs == 0) {
      return SILO_GPU_OK;
   }
   uint8_t* d_membership = nullptr;
   SILO_HIP_TRY(hipMalloc(&d_membership, n_ids));
   hipError_t err = hipMemcpyAsync(d_membership, membership_by_id, n_ids, hipMemcpyHostToDevice, hip_stream);
   if (err == hipSuccess) {
      k_bitset_from_pairs<<<(n_pairs + 255) / 256, 256, 0, hip_stream>>>(rows_dev, ids_dev, n_pairs, d_membership, dst_dev);
      err = hipStreamSynchronize(hip_stream);  // the membership table is freed below
   }
   (void)hipFree(d_membership);
   if (err != hipSuccess) {
      (void)hipGetLastError();
      return silo_gpu_internal_fail(SILO_GPU_ERR_HIP, std::string("k_bitset_from_pairs: ") + hipGetErrorString(err));
   }
   return SILO_GPU_OK;
}

int silo_gpu_count_pairs(
   const silo_gpu_store* store, const uint64_t* filter_dev, const uint32_t* rows_dev, const uint32_t* ids_dev, uint32_t n_pairs,
   uint32_t* counts_dev, void* stream
) {
   if (store == nullptr || counts_dev == nullptr || (n_pairs != 0 && (rows_dev == nullptr || ids_dev == nullptr))) {
      return silo_gpu_internal_fail(SILO_GPU_ERR_INVALID_ARGUMENT, "silo_gpu_count_pairs: bad arguments");
   }
   if (n_pairs == 0) {
      return SILO_GPU_OK;
   }
   k_count_pairs<<<(n_pairs + 255) / 256, 256, 0, static_cast<hipStream_t>(stream)>>>(filter_dev, rows_dev, ids_dev, n_pairs, counts_dev);
   SILO_HIP_TRY(hipGetLastError());
   return SILO_GPU_OK;
}

}  // extern "C"
