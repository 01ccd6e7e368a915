// bitprog.h — one-word interpreter of the fused filter bit-program (K3).
//
// The device kernel (silo_gpu.hip: k_filter_eval) runs this once per 64-bit bitset word with the
// slots in LDS; tests/host_tools/bitprog_host.cpp compiles the very same function with g++ so the
// instruction semantics can be unit-tested on a machine without a GPU.  It is not a CPU fallback:
// nothing in the product path calls the host build.
//
// Semantics follow the reference operators (file:line under the reference tree):
//   NOT      operators/complement.cpp:50-54     flip(0,row_count)  ->  ~x & valid
//   AND      operators/intersection.cpp:111-126 &=
//   ANDNOT   operators/intersection.cpp:115,124 -=
//   OR       operators/union.cpp:44             fastunion
//   CNT_*    operators/threshold.cpp:64-138     the n-of-k DP table, restated as a bit-sliced
//            vertical counter: CNT_ADD ripple-adds one child plane, CNT_GE / CNT_EQ compare the
//            per-row count with n (exact = table[n-1] - table[n], threshold.cpp:130-135).
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define SILO_HD __host__ __device__ __forceinline__
#else
#define SILO_HD inline
#endif

#include "../../include/silo_gpu.h"

namespace silo_gpu {

// `slot(i)` returns a uint64_t& ; `leaf(i)` returns the word of leaf i.
template <class SlotFn, class LeafFn>
SILO_HD uint64_t bitprog_run_word(
   const uint32_t* code, uint32_t n_instructions, uint64_t valid, SlotFn slot, LeafFn leaf
) {
   for (uint32_t pc = 0; pc < n_instructions; ++pc) {
      const uint32_t w0 = code[2 * pc];
      const uint32_t imm = code[2 * pc + 1];
      const uint32_t op = w0 & 0xFFu;
      const uint32_t dst = (w0 >> 8) & 0xFFu;
      const uint32_t a = (w0 >> 16) & 0xFFu;
      const uint32_t b = (w0 >> 24) & 0xFFu;
      switch (op) {
         case SILO_GPU_OP_LOAD:
            slot(dst) = leaf(imm);
            break;
         case SILO_GPU_OP_ZERO:
            slot(dst) = 0;
            break;
         case SILO_GPU_OP_ONES:
            slot(dst) = valid;
            break;
         case SILO_GPU_OP_NOT:
            slot(dst) = ~slot(a) & valid;
            break;
         case SILO_GPU_OP_AND:
            slot(dst) = slot(a) & slot(b);
            break;
         case SILO_GPU_OP_OR:
            slot(dst) = slot(a) | slot(b);
            break;
         case SILO_GPU_OP_ANDNOT:
            slot(dst) = slot(a) & ~slot(b);
            break;
         case SILO_GPU_OP_MOV:
            slot(dst) = slot(a);
            break;
         case SILO_GPU_OP_CNT_ADD: {
            // counter bits live in slots dst .. dst+b-1 (LSB first); add the 1-bit plane slot(a)
            uint64_t carry = slot(a);
            for (uint32_t bit = 0; bit < b; ++bit) {
               const uint64_t cur = slot(dst + bit);
               slot(dst + bit) = cur ^ carry;
               carry &= cur;
            }
            break;
         }
         case SILO_GPU_OP_CNT_GE:
         case SILO_GPU_OP_CNT_EQ: {
            // compare the b-bit counter in slots a .. a+b-1 with imm, MSB first
            uint64_t greater = 0;
            uint64_t equal = ~0ull;
            if (b < 32 && (imm >> b) != 0) {
               equal = 0;  // imm does not fit in b bits: counter < imm everywhere
            } else {
               for (int bit = static_cast<int>(b) - 1; bit >= 0; --bit) {
                  const uint64_t cur = slot(a + static_cast<uint32_t>(bit));
                  if ((imm >> bit) & 1u) {
                     equal &= cur;
                  } else {
                     greater |= equal & cur;
                     equal &= ~cur;
                  }
               }
            }
            slot(dst) = (op == SILO_GPU_OP_CNT_GE ? (greater | equal) : equal) & valid;
            break;
         }
         default:
            break;
      }
   }
   return slot(0);
}

// valid(w): rows 64w .. 64w+63 that are < sequence_count
SILO_HD uint64_t valid_mask(uint32_t word, uint32_t sequence_count) {
   const uint64_t first = static_cast<uint64_t>(word) * 64u;
   if (first >= sequence_count) {
      return 0;
   }
   const uint64_t remaining = sequence_count - first;
   return remaining >= 64 ? ~0ull : ((1ull << remaining) - 1ull);
}

}  // namespace silo_gpu
