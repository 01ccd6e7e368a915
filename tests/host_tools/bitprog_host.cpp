// g++ build of the bit-program interpreter (lapis-silo_amd/csrc/bitprog.h) for CPU unit tests of the
// host logic.  Test tool only: nothing in the product links or calls it.
#include <cstdint>
#include <vector>

#include "../../lapis-silo_amd/csrc/bitprog.h"

extern "C" int bitprog_eval_host(
   const uint32_t* code, uint32_t n_instructions, const uint64_t* leaves /* [n_leaves][n_words] */, uint32_t n_leaves,
   uint32_t n_words, uint32_t sequence_count, uint64_t* out /* [n_words] */
) {
   std::vector<uint64_t> slots(SILO_GPU_MAX_SLOTS);
   for (uint32_t w = 0; w < n_words; ++w) {
      const uint64_t valid = silo_gpu::valid_mask(w, sequence_count);
      const auto leaf = [&](uint32_t index) -> uint64_t {
         return index < n_leaves ? leaves[static_cast<size_t>(index) * n_words + w] : 0;
      };
      const auto get = [&](uint32_t index) -> uint64_t {
         return index >= SILO_GPU_LEAF_OPERAND ? leaf(index - SILO_GPU_LEAF_OPERAND) : slots[index];
      };
      const auto set = [&](uint32_t index, uint64_t value) { slots[index] = value; };
      out[w] = silo_gpu::bitprog_run<uint64_t>(code, n_instructions, valid, get, set, leaf) & valid;
   }
   return 0;
}
