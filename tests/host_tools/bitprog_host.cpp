// g++ build of the bit-program interpreter (lapis-silo_amd/csrc/bitprog.h) for CPU unit tests of the
// host logic.  Test tool only: nothing in the product links or calls it.
#include <cstdint>
#include <vector>

#include "../../lapis-silo_amd/csrc/bitprog.h"

extern "C" int bitprog_eval_host(
   const uint32_t* code, uint32_t n_instructions, const uint64_t* leaves /* [n_leaves][n_words] */, uint32_t n_leaves,
   uint32_t n_words, uint32_t sequence_count, uint64_t* out /* [n_words] */
) {
   std::vector<uint64_t> slots(SILO_GPU_LEAF_OPERAND + SILO_GPU_MAX_LEAVES);
   for (uint32_t w = 0; w < n_words; ++w) {
      const uint64_t valid = silo_gpu::valid_mask(w, sequence_count);
      for (uint32_t leaf = 0; leaf < n_leaves; ++leaf) {  // operands >= SILO_GPU_LEAF_OPERAND read a leaf
         slots[SILO_GPU_LEAF_OPERAND + leaf] = leaves[static_cast<size_t>(leaf) * n_words + w];
      }
      out[w] = silo_gpu::bitprog_run_word(
                  code, n_instructions, valid, [&](uint32_t slot) -> uint64_t& { return slots[slot]; },
                  [&](uint32_t leaf) -> uint64_t { return leaf < n_leaves ? leaves[static_cast<size_t>(leaf) * n_words + w] : 0; }
               ) &
               valid;
   }
   return 0;
}
