// silo_gpu_columns.hip — metadata columns on the device.
//
//   K5  k_bitset_from_compare   a column predicate (CompareToValueSelection<T>::match, selection.cpp:145-165) for
//                               all rows at once: bit i = values[i] <op> value; one wave ballot per bitset word
//   K6  k_group_count           Aggregated with groupByFields (aggregated.cpp:100-149): histogram of the combined
//                               dictionary ids of the filtered rows
//
// Both stream a 4- or 8-byte column once (HBM-bound, 4..8 B per row) and write 1 bit / a few counters per row.
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

// One wave per bitset word: lane l tests row 64 * word + l (a coalesced 256- or 512-byte read per wave), the
// ballot is the word.  Words past the last row (row padding) are written as zero.
template <typename T>
__global__ __launch_bounds__(256) void k_bitset_from_compare(
   const T* __restrict__ values, uint32_t n_rows, uint32_t row_words, int comparator, T value, uint64_t* __restrict__ out
) {
   const uint32_t word = blockIdx.x * (blockDim.x / 64) + (threadIdx.x >> 6);
   if (word >= row_words) {
      return;
   }
   const uint32_t row = word * 64 + (threadIdx.x & 63u);
   const bool match = row < n_rows && compareValues<T>(values[row], comparator, value);
   const uint64_t ballot = __ballot(match);
   if ((threadIdx.x & 63u) == 0) {
      out[word] = ballot;
   }
}

constexpr uint32_t GROUP_LDS_BINS = 4096;
constexpr uint32_t GROUP_ROWS_PER_THREAD = 16;

struct GroupCountArgs {
   const uint32_t* ids[SILO_GPU_MAX_GROUP_COLUMNS];
   uint32_t strides[SILO_GPU_MAX_GROUP_COLUMNS];
   uint32_t n_columns;
};

// Adds one to bins[key] for every active lane; lanes that share the key of the first active lane are counted by
// one atomic (a few rounds of that take the sting out of skewed columns, where most rows fall into one group).
template <typename Add>
__device__ __forceinline__ void aggregatedIncrement(bool active, uint32_t key, Add add) {
#pragma unroll 1
   for (int round = 0; round < 4; ++round) {
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
      }
      active = active && !same;
   }
   if (active) {
      add(key, 1u);
   }
}

template <bool USE_LDS>
__global__ __launch_bounds__(256) void k_group_count(
   const uint64_t* __restrict__ filter, uint32_t n_rows, const GroupCountArgs args, uint32_t n_bins, uint32_t* __restrict__ counts
) {
   __shared__ uint32_t s_bins[USE_LDS ? GROUP_LDS_BINS : 1];
   if (USE_LDS) {
      for (uint32_t bin = threadIdx.x; bin < n_bins; bin += blockDim.x) {
         s_bins[bin] = 0;
      }
      __syncthreads();
   }
   const uint32_t block_first = blockIdx.x * (blockDim.x * GROUP_ROWS_PER_THREAD);
#pragma unroll 1
   for (uint32_t step = 0; step < GROUP_ROWS_PER_THREAD; ++step) {
      const uint32_t row = block_first + step * blockDim.x + threadIdx.x;  // consecutive lanes, consecutive rows
      bool active = row < n_rows;
      if (active && filter != nullptr) {
         active = (filter[row >> 6] >> (row & 63u)) & 1u;
      }
      uint32_t key = 0;
      if (active) {
         for (uint32_t column = 0; column < args.n_columns; ++column) {
            key += args.ids[column][row] * args.strides[column];
         }
      }
      if (USE_LDS) {
         aggregatedIncrement(active, key, [&](uint32_t bin, uint32_t n) { atomicAdd(&s_bins[bin], n); });
      } else {
         aggregatedIncrement(active, key, [&](uint32_t bin, uint32_t n) { atomicAdd(&counts[bin], n); });
      }
   }
   if (USE_LDS) {
      __syncthreads();
      for (uint32_t bin = threadIdx.x; bin < n_bins; bin += blockDim.x) {
         const uint32_t n = s_bins[bin];
         if (n != 0) {
            atomicAdd(&counts[bin], n);
         }
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
   const dim3 grid((row_words + 3) / 4);
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
   const uint32_t rows_per_block = 256 * GROUP_ROWS_PER_THREAD;
   const dim3 grid((n_rows + rows_per_block - 1) / rows_per_block);
   if (n_bins <= GROUP_LDS_BINS) {
      k_group_count<true><<<grid, 256, 0, hip_stream>>>(filter_dev, n_rows, args, static_cast<uint32_t>(n_bins), counts_dev);
   } else {
      k_group_count<false><<<grid, 256, 0, hip_stream>>>(filter_dev, n_rows, args, static_cast<uint32_t>(n_bins), counts_dev);
   }
   SILO_HIP_TRY(hipGetLastError());
   return SILO_GPU_OK;
}

}  // extern "C"
