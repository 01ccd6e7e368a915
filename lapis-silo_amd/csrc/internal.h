// internal.h — what the translation units of libsilo_gpu.so share (not part of the C ABI).
#pragma once
#include <hip/hip_runtime.h>

#include <string>

#include "silo_gpu.h"

/// Records `message` as this thread's silo_gpu_last_error() and returns `code`.
int silo_gpu_internal_fail(int code, const std::string& message);

/// Sorts n 64-bit keys in device memory ascending, in place (silo_gpu_sort.hip); synchronises the null stream.
int silo_gpu_internal_sort_keys(uint64_t* keys_dev, size_t n);

/// The same by the key bits [begin_bit, end_bit) only; stable (keys equal in those bits keep their order).
int silo_gpu_internal_sort_keys_by_bits(uint64_t* keys_dev, size_t n, int begin_bit, int end_bit);

/// Sorts n (64-bit key, 32-bit value) pairs in device memory by key, ascending, in place; synchronises the null stream.
int silo_gpu_internal_sort_pairs(uint64_t* keys_dev, uint32_t* values_dev, size_t n);

#define SILO_HIP_TRY(expr)                                                                                  \
   do {                                                                                                     \
      hipError_t err_ = (expr);                                                                             \
      if (err_ != hipSuccess) {                                                                             \
         (void)hipGetLastError(); /* clear the sticky error so later launch checks start clean */           \
         return silo_gpu_internal_fail(                                                                     \
            err_ == hipErrorOutOfMemory ? SILO_GPU_ERR_OUT_OF_MEMORY : SILO_GPU_ERR_HIP,                    \
            std::string(#expr) + ": " + hipGetErrorString(err_)                                             \
         );                                                                                                 \
      }                                                                                                     \
   } while (0)
