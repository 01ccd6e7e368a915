// silo_gpu_runtime.hip — errors, tuning knobs, memory / stream / event wrappers of the C ABI, the stream-read probe.
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

std::atomic<int> g_tune_rows_per_block{0};
std::atomic<int> g_tune_scan_variant{0};
std::atomic<int> g_tune_eval_leaf_batch{0};
std::atomic<int> g_tune_compact_index{0};
std::atomic<int> g_tune_side_stream{0};
std::atomic<int> g_tune_scan_timing{0};
std::atomic<int> g_tune_missing_runs{0};
std::atomic<int> g_tune_key_cost{0};
std::atomic<int> g_tune_launch_cost{0};
std::atomic<int> g_tune_sparse_divisor{0};

namespace {
thread_local std::string g_last_error;
}  // namespace

int silo_gpu_internal_fail(int code, const std::string& message) {  // for every translation unit (internal.h)
   g_last_error = message;
   return code;
}

using namespace silo_gpu_detail;

extern "C" {

const char* silo_gpu_last_error(void) {
   return g_last_error.c_str();
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
