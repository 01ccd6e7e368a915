// silo_gpu_comm.hip — the exchange step of the sharded Mutations scan: RCCL over xGMI behind the C ABI.
//
// position-range shards all-reduce (sum, uint32) the count table counts[P][S] once per query
// (0.6 MB for the nucleotide genome: latency-bound, one collective, no bucketing); sequence-id shards do the same with
// their partial counts; a filter leaf at a position another rank owns travels as ONE broadcast of a row bitset.
// The reference has no analogue (one process, roaring bitmaps in host memory: query_engine.cpp:40-49 sums its
// partitions in a loop) — this is what stands where that loop crosses GPUs.
//
// librccl is bound at run time (dlopen), not at link time: a single-GPU engine never maps the 570 MB library, and a
// process that already carries an RCCL (a PyTorch host loads its own copy under the same SONAME) keeps using that one.
// A communicator is NOT safe for concurrent enqueues, and every rank has to enqueue the collectives in the same order:
// calls on one communicator are serialised by its mutex, and the sharded engine runs its queries SPMD (same queries,
// same order on every rank).
#include <dlfcn.h>
#include <rccl/rccl.h>
#include <string.h>

#include <mutex>
#include <new>
#include <string>
#include <type_traits>

#include "internal.h"

namespace {

struct RcclApi {
   void* handle = nullptr;
   ncclResult_t (*get_unique_id)(ncclUniqueId*) = nullptr;
   ncclResult_t (*comm_init_rank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
   ncclResult_t (*comm_destroy)(ncclComm_t) = nullptr;
   ncclResult_t (*all_reduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
   ncclResult_t (*broadcast)(const void*, void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
   const char* (*get_error_string)(ncclResult_t) = nullptr;
   std::string error;
};

RcclApi& rccl() {
   static RcclApi api;
   static std::once_flag once;
   std::call_once(once, [] {
      // an RCCL that is already mapped (same SONAME) wins; otherwise the ROCm installation's
      const char* candidates[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
      api.handle = dlopen(candidates[0], RTLD_NOW | RTLD_NOLOAD);
      for (const char* name : candidates) {
         if (api.handle != nullptr) {
            break;
         }
         api.handle = dlopen(name, RTLD_NOW | RTLD_LOCAL);
      }
      if (api.handle == nullptr) {
         const char* reason = dlerror();
         api.error = std::string("librccl could not be loaded: ") + (reason != nullptr ? reason : "unknown reason");
         return;
      }
      const auto bind = [&](const char* symbol, auto& target) {
         target = reinterpret_cast<std::decay_t<decltype(target)>>(dlsym(api.handle, symbol));
         if (target == nullptr && api.error.empty()) {
            api.error = std::string("librccl lacks ") + symbol;
         }
      };
      bind("ncclGetUniqueId", api.get_unique_id);
      bind("ncclCommInitRank", api.comm_init_rank);
      bind("ncclCommDestroy", api.comm_destroy);
      bind("ncclAllReduce", api.all_reduce);
      bind("ncclBroadcast", api.broadcast);
      bind("ncclGetErrorString", api.get_error_string);
   });
   return api;
}

int rcclReady() {
   const RcclApi& api = rccl();
   return api.error.empty() ? SILO_GPU_OK : silo_gpu_internal_fail(SILO_GPU_ERR_UNSUPPORTED, api.error);
}

int rcclFail(const char* what, ncclResult_t status) {
   return silo_gpu_internal_fail(SILO_GPU_ERR_HIP, std::string(what) + ": " + rccl().get_error_string(status));
}

}  // namespace

struct silo_gpu_comm {
   ncclComm_t comm = nullptr;
   int device = 0;
   uint32_t rank = 0;
   uint32_t world = 1;
   std::mutex mutex;  // one enqueue at a time
};

extern "C" {

int silo_gpu_comm_unique_id(uint8_t* out_id) {
   static_assert(SILO_GPU_COMM_ID_BYTES == NCCL_UNIQUE_ID_BYTES, "the id travels as opaque bytes");
   if (out_id == nullptr) {
      return silo_gpu_internal_fail(SILO_GPU_ERR_INVALID_ARGUMENT, "silo_gpu_comm_unique_id: null out pointer");
   }
   if (const int rc = rcclReady(); rc != SILO_GPU_OK) {
      return rc;
   }
   ncclUniqueId id;
   if (const ncclResult_t status = rccl().get_unique_id(&id); status != ncclSuccess) {
      return rcclFail("ncclGetUniqueId", status);
   }
   memcpy(out_id, id.internal, NCCL_UNIQUE_ID_BYTES);
   return SILO_GPU_OK;
}

int silo_gpu_comm_create(const uint8_t* id, uint32_t rank, uint32_t world, int device, silo_gpu_comm** out) {
   if (id == nullptr || out == nullptr || world == 0 || rank >= world) {
      return silo_gpu_internal_fail(SILO_GPU_ERR_INVALID_ARGUMENT, "silo_gpu_comm_create: bad arguments");
   }
   *out = nullptr;
   if (const int rc = rcclReady(); rc != SILO_GPU_OK) {
      return rc;
   }
   int device_count = 0;
   if (hipGetDeviceCount(&device_count) != hipSuccess || device_count == 0) {
      (void)hipGetLastError();
      return silo_gpu_internal_fail(SILO_GPU_ERR_NO_DEVICE, "silo_gpu_comm_create: no HIP device visible");
   }
   if (device < 0 || device >= device_count) {
      return silo_gpu_internal_fail(SILO_GPU_ERR_INVALID_ARGUMENT, "silo_gpu_comm_create: device ordinal out of range");
   }
   SILO_HIP_TRY(hipSetDevice(device));
   auto* comm = new (std::nothrow) silo_gpu_comm;
   if (comm == nullptr) {
      return silo_gpu_internal_fail(SILO_GPU_ERR_OUT_OF_MEMORY, "out of host memory");
   }
   comm->device = device;
   comm->rank = rank;
   comm->world = world;
   ncclUniqueId unique;
   memcpy(unique.internal, id, NCCL_UNIQUE_ID_BYTES);
   if (const ncclResult_t status = rccl().comm_init_rank(&comm->comm, static_cast<int>(world), unique, static_cast<int>(rank)); status != ncclSuccess) {
      delete comm;
      return rcclFail("ncclCommInitRank", status);
   }
   *out = comm;
   return SILO_GPU_OK;
}

void silo_gpu_comm_destroy(silo_gpu_comm* comm) {
   if (comm == nullptr) {
      return;
   }
   if (comm->comm != nullptr) {
      (void)hipSetDevice(comm->device);
      (void)rccl().comm_destroy(comm->comm);
   }
   delete comm;
}

uint32_t silo_gpu_comm_rank(const silo_gpu_comm* comm) {
   return comm != nullptr ? comm->rank : 0;
}

uint32_t silo_gpu_comm_world(const silo_gpu_comm* comm) {
   return comm != nullptr ? comm->world : 0;
}

int silo_gpu_allreduce_counts(silo_gpu_comm* comm, uint32_t* counts_dev, size_t n, void* stream) {
   if (comm == nullptr || (counts_dev == nullptr && n != 0)) {
      return silo_gpu_internal_fail(SILO_GPU_ERR_INVALID_ARGUMENT, "silo_gpu_allreduce_counts: bad arguments");
   }
   if (n == 0) {
      return SILO_GPU_OK;
   }
   const std::lock_guard<std::mutex> lock(comm->mutex);
   SILO_HIP_TRY(hipSetDevice(comm->device));
   const ncclResult_t status = rccl().all_reduce(counts_dev, counts_dev, n, ncclUint32, ncclSum, comm->comm, static_cast<hipStream_t>(stream));
   return status == ncclSuccess ? SILO_GPU_OK : rcclFail("ncclAllReduce", status);
}

int silo_gpu_broadcast_bytes(silo_gpu_comm* comm, void* bytes_dev, size_t n_bytes, uint32_t root, void* stream) {
   if (comm == nullptr || (bytes_dev == nullptr && n_bytes != 0) || root >= comm->world) {
      return silo_gpu_internal_fail(SILO_GPU_ERR_INVALID_ARGUMENT, "silo_gpu_broadcast_bytes: bad arguments");
   }
   if (n_bytes == 0) {
      return SILO_GPU_OK;
   }
   const std::lock_guard<std::mutex> lock(comm->mutex);
   SILO_HIP_TRY(hipSetDevice(comm->device));
   const ncclResult_t status =
      rccl().broadcast(bytes_dev, bytes_dev, n_bytes, ncclUint8, static_cast<int>(root), comm->comm, static_cast<hipStream_t>(stream));
   return status == ncclSuccess ? SILO_GPU_OK : rcclFail("ncclBroadcast", status);
}

}  // extern "C"
