// silo_gpu_sort.hip — device radix sort of 64-bit keys (rocPRIM through hipCUB), kept in its own translation unit so
// that the kernel file does not pay for the template instantiation on every rebuild.  Used once per sequence store at
// finalize: the escape keys position << 37 | symbol << 32 | sequence of the adaptive code planes are put in ascending
// order, which groups them by (position, symbol) for the scan's escape pass and makes a (position, symbol, sequence)
// lookup a binary search (FastaAligned); a second, STABLE pass over a few sequence bits then gives the slice-major copy of
// the keys that the scan's escape pass streams (keys of one slice of the rows together, (position, symbol, sequence) within).
#include <hipcub/hipcub.hpp>

#include "internal.h"

int silo_gpu_internal_sort_keys(uint64_t* keys_dev, size_t n) {
   return silo_gpu_internal_sort_keys_by_bits(keys_dev, n, 0, 64);
}

int silo_gpu_internal_sort_keys_by_bits(uint64_t* keys_dev, size_t n, int begin_bit, int end_bit) {
   if (n < 2 || begin_bit >= end_bit) {
      return SILO_GPU_OK;
   }
   uint64_t* sorted = nullptr;
   void* scratch = nullptr;
   size_t scratch_bytes = 0;
   SILO_HIP_TRY(hipMalloc(&sorted, n * sizeof(uint64_t)));
   hipError_t status = hipcub::DeviceRadixSort::SortKeys(nullptr, scratch_bytes, keys_dev, sorted, n, begin_bit, end_bit, nullptr);
   if (status == hipSuccess) {
      status = hipMalloc(&scratch, scratch_bytes);
   }
   if (status == hipSuccess) {
      status = hipcub::DeviceRadixSort::SortKeys(scratch, scratch_bytes, keys_dev, sorted, n, begin_bit, end_bit, nullptr);
   }
   if (status == hipSuccess) {
      status = hipMemcpy(keys_dev, sorted, n * sizeof(uint64_t), hipMemcpyDeviceToDevice);
   }
   (void)hipFree(scratch);
   (void)hipFree(sorted);
   SILO_HIP_TRY(status);
   return SILO_GPU_OK;
}

int silo_gpu_internal_sort_pairs(uint64_t* keys_dev, uint32_t* values_dev, size_t n) {
   if (n < 2) {
      return SILO_GPU_OK;
   }
   uint64_t* sorted_keys = nullptr;
   uint32_t* sorted_values = nullptr;
   void* scratch = nullptr;
   size_t scratch_bytes = 0;
   hipError_t status = hipMalloc(&sorted_keys, n * sizeof(uint64_t));
   if (status == hipSuccess) {
      status = hipMalloc(&sorted_values, n * sizeof(uint32_t));
   }
   if (status == hipSuccess) {
      status = hipcub::DeviceRadixSort::SortPairs(nullptr, scratch_bytes, keys_dev, sorted_keys, values_dev, sorted_values, n, 0, 64, nullptr);
   }
   if (status == hipSuccess) {
      status = hipMalloc(&scratch, scratch_bytes);
   }
   if (status == hipSuccess) {
      status = hipcub::DeviceRadixSort::SortPairs(scratch, scratch_bytes, keys_dev, sorted_keys, values_dev, sorted_values, n, 0, 64, nullptr);
   }
   if (status == hipSuccess) {
      status = hipMemcpy(keys_dev, sorted_keys, n * sizeof(uint64_t), hipMemcpyDeviceToDevice);
   }
   if (status == hipSuccess) {
      status = hipMemcpy(values_dev, sorted_values, n * sizeof(uint32_t), hipMemcpyDeviceToDevice);
   }
   (void)hipFree(scratch);
   (void)hipFree(sorted_keys);
   (void)hipFree(sorted_values);
   SILO_HIP_TRY(status);
   return SILO_GPU_OK;
}
