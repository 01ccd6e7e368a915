// bitprog.h — interpreter of the fused filter bit-program (K3), shared by the device kernel and a g++ build.
//
// The device kernel (silo_gpu.hip: k_filter_eval) runs this once per lane with T = two 64-bit bitset
// words and the slots in LDS; tests/host_tools/bitprog_host.cpp compiles the very same function with
// g++ (T = one word) so the instruction semantics can be unit-tested on a machine without a GPU.  It is
// not a CPU fallback: nothing in the product path calls the host build.
//
// Semantics follow the reference operators (file:line under the reference tree):
//   NOT           operators/complement.cpp:50-54     flip(0,row_count)  ->  ~x & valid
//   AND / AND_N   operators/intersection.cpp:111-126 &=
//   ANDNOT        operators/intersection.cpp:115,124 -=
//   OR / OR_N     operators/union.cpp:44             fastunion
//   CNT_*         operators/threshold.cpp:64-138     the n-of-k DP table, restated as a bit-sliced
//                 vertical counter: CNT_ADD ripple-adds one child plane, CNT_GE / CNT_EQ compare the
//                 per-row count with n (exact = table[n-1] - table[n], threshold.cpp:130-135).
// The *_N forms take a run of consecutive leaves (imm = first | count << 16) and fetch them 8 at a time
// with independent loads, so a flat Or / And / N-Of over stored columns streams at memory speed instead
// of paying one dependent load per LOAD instruction.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define SILO_HD __host__ __device__ __forceinline__
#else
#define SILO_HD inline
#endif

#include "../../include/silo_gpu.h"

namespace silo_gpu {

// Two bitset words handled by one lane (one 16-byte load per leaf).
struct Word2 {
   uint64_t x, y;
};
SILO_HD Word2 operator&(Word2 a, Word2 b) { return {a.x & b.x, a.y & b.y}; }
SILO_HD Word2 operator|(Word2 a, Word2 b) { return {a.x | b.x, a.y | b.y}; }
SILO_HD Word2 operator^(Word2 a, Word2 b) { return {a.x ^ b.x, a.y ^ b.y}; }
SILO_HD Word2 operator~(Word2 a) { return {~a.x, ~a.y}; }

template <class T>
SILO_HD T zeroOf();
template <>
SILO_HD uint64_t zeroOf<uint64_t>() { return 0; }
template <>
SILO_HD Word2 zeroOf<Word2>() { return {0, 0}; }

constexpr uint32_t LEAF_BATCH = 8;

// get(index): value of slot `index`, or of leaf (index - SILO_GPU_LEAF_OPERAND) for index >= SILO_GPU_LEAF_OPERAND
// set(index, value): store into slot `index`        leaf(i): value of leaf i (a global load on the device)
template <class T, uint32_t BATCH = LEAF_BATCH, class GetFn, class SetFn, class LeafFn>
SILO_HD T bitprog_run(const uint32_t* code, uint32_t n_instructions, T valid, GetFn get, SetFn set, LeafFn leaf) {
   const T ones = ~zeroOf<T>();
   for (uint32_t pc = 0; pc < n_instructions; ++pc) {
      const uint32_t w0 = code[2 * pc];
      const uint32_t imm = code[2 * pc + 1];
      const uint32_t op = w0 & 0xFFu;
      const uint32_t dst = (w0 >> 8) & 0xFFu;
      const uint32_t a = (w0 >> 16) & 0xFFu;
      const uint32_t b = (w0 >> 24) & 0xFFu;
      switch (op) {
         case SILO_GPU_OP_LOAD:
            set(dst, leaf(imm));
            break;
         case SILO_GPU_OP_ZERO:
            set(dst, zeroOf<T>());
            break;
         case SILO_GPU_OP_ONES:
            set(dst, valid);
            break;
         case SILO_GPU_OP_NOT:
            set(dst, ~get(a) & valid);
            break;
         case SILO_GPU_OP_AND:
            set(dst, get(a) & get(b));
            break;
         case SILO_GPU_OP_OR:
            set(dst, get(a) | get(b));
            break;
         case SILO_GPU_OP_ANDNOT:
            set(dst, get(a) & ~get(b));
            break;
         case SILO_GPU_OP_MOV:
            set(dst, get(a));
            break;
         case SILO_GPU_OP_OR_N:
         case SILO_GPU_OP_AND_N: {
            const uint32_t first = imm & 0xFFFFu;
            const uint32_t count = imm >> 16;
            const uint32_t last = first + count - 1;
            T acc = op == SILO_GPU_OP_OR_N ? zeroOf<T>() : ones;
            for (uint32_t i = 0; i < count; i += BATCH) {
               T value[BATCH];
#pragma unroll
               for (uint32_t k = 0; k < BATCH; ++k) {  // clamped: the last leaf is re-read, never a branch around a load
                  const uint32_t index = first + i + k;
                  value[k] = leaf(index < last ? index : last);
               }
#pragma unroll
               for (uint32_t k = 0; k < BATCH; ++k) {
                  acc = op == SILO_GPU_OP_OR_N ? (acc | value[k]) : (acc & value[k]);
               }
            }
            set(dst, acc);
            break;
         }
         case SILO_GPU_OP_CNT_ADD: {
            // counter bits live in slots dst .. dst+b-1 (LSB first); add the 1-bit plane get(a)
            T carry = get(a);
            for (uint32_t bit = 0; bit < b; ++bit) {
               const T cur = get(dst + bit);
               set(dst + bit, cur ^ carry);
               carry = carry & cur;
            }
            break;
         }
         case SILO_GPU_OP_CNT_ADD_N:
         case SILO_GPU_OP_CNT_ADD_NOT_N: {
            // add every leaf of the run (or its complement within `valid`) to the counter dst .. dst+b-1
            const uint32_t first = imm & 0xFFFFu;
            const uint32_t count = imm >> 16;
            for (uint32_t i = 0; i < count; i += BATCH) {
               T value[BATCH];
#pragma unroll
               for (uint32_t k = 0; k < BATCH; ++k) {
                  const uint32_t index = first + i + k;
                  value[k] = leaf(index < first + count ? index : first + count - 1);
               }
#pragma unroll
               for (uint32_t k = 0; k < BATCH; ++k) {  // static index into value[]: stays in registers
                  if (i + k < count) {
                     T carry = op == SILO_GPU_OP_CNT_ADD_N ? value[k] : (~value[k] & valid);
                     for (uint32_t bit = 0; bit < b; ++bit) {
                        const T cur = get(dst + bit);
                        set(dst + bit, cur ^ carry);
                        carry = carry & cur;
                     }
                  }
               }
            }
            break;
         }
         case SILO_GPU_OP_CNT_GE:
         case SILO_GPU_OP_CNT_EQ: {
            // compare the b-bit counter in slots a .. a+b-1 with imm, MSB first
            T greater = zeroOf<T>();
            T equal = ones;
            if (b < 32 && (imm >> b) != 0) {
               equal = zeroOf<T>();  // imm does not fit in b bits: counter < imm everywhere
            } else {
               for (int bit = static_cast<int>(b) - 1; bit >= 0; --bit) {
                  const T cur = get(a + static_cast<uint32_t>(bit));
                  if ((imm >> bit) & 1u) {
                     equal = equal & cur;
                  } else {
                     greater = greater | (equal & cur);
                     equal = equal & ~cur;
                  }
               }
            }
            set(dst, (op == SILO_GPU_OP_CNT_GE ? (greater | equal) : equal) & valid);
            break;
         }
         default:
            break;
      }
   }
   return get(0);
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
