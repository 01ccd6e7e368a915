// silo_gpu_import.hip — import of the reference's own storage form: the roaring payloads of a snapshot.
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
   uint32_t* sparse_count, uint32_t sparse_capacity, uint32_t* overlap_flag
) {
   const uint32_t w = blockIdx.x * blockDim.x + threadIdx.x;
   if (w >= store.row_words) {
      return;
   }
   const uint64_t bits = row[w];
   if ((seen[w] & bits) != 0) {  // a row that has a symbol at this position already (overlapping bitmaps, a position imported twice):
      atomicOr(overlap_flag, 1u);  // refused — OR-ing two symbols' code bits would give the row the code of a third
      return;
   }
   seen[w] |= bits;
   if (bits == 0) {
      return;
   }
   const uint8_t kind = store.kind[symbol];
   if (kind == PLANE_SCAN) {
      const uint32_t code = static_cast<uint32_t>(store.index[symbol]) + 1u;
      uint64_t* planes = store.scan + static_cast<size_t>(position) * store.n_bits * store.row_words + w;
      uint64_t coded = 0;  // rows of this word that carry a code at the position already: the position was imported before
      for (uint32_t bit = 0; bit < store.n_bits; ++bit) {
         coded |= planes[static_cast<size_t>(bit) * store.row_words];
      }
      if ((coded & bits) != 0) {
         atomicOr(overlap_flag, 1u);
         return;
      }
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
         seqstore.dev, position, symbol, store->d_import_row, store->d_import_union, seqstore.d_sparse, seqstore.d_sparse_count, seqstore.sparse_capacity,
         store->d_error_flag
      );
      HIP_TRY(hipDeviceSynchronize());
      HIP_TRY(hipMemcpy(&count_before, seqstore.d_sparse_count, sizeof(uint32_t), hipMemcpyDeviceToHost));
      uint32_t overlap = 0;
      HIP_TRY(hipMemcpy(&overlap, store->d_error_flag, sizeof(uint32_t), hipMemcpyDeviceToHost));
      if (overlap != 0) {
         HIP_TRY(hipMemset(store->d_error_flag, 0, sizeof(uint32_t)));
         return fail(SILO_GPU_ERR_INVALID_ARGUMENT, "silo_gpu_store_import_position: the bitmaps of the position overlap (a row with two symbols) — or the position was imported before");
      }
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
