// layout_choice.h — the layout of every position of a finalized sequence store (host code, shared by silo_gpu.hip and the
// CPU unit tests: no device is involved).  See "The adaptive code planes" in silo_gpu.hip and DESIGN.md §2.
#pragma once
#include <stdint.h>

#include <algorithm>
#include <vector>

namespace silo_gpu_layout {

// code_map[p][0]: the number of plane rows of the position, with
constexpr uint8_t LAYOUT_IDENTITY = 0x80;  // identity code planes (code = index of the valid mutation symbol + 1, no escapes), or
constexpr uint8_t LAYOUT_ONE_HOT = 0x40;   // ONE-HOT rows: k = 1..3 rows, row j = the rows of the symbol code_map[p][1 + j]
constexpr uint32_t CODE_MAP_STRIDE = 8;    // bytes of code_map per position: [0] = layout, [c] = scan symbol of code c (1..7), 0xFF = unused
// what an escape key costs a scan, in plane bytes (the escape pass streams its 8-byte keys beside the plane scans and shares
// the HBM with them; the optimum is flat between 10 and 24 — profiles/r02_one_hot_rows.md)
constexpr uint32_t KEY_COST_BYTES = 16;

/// The layout of every position of a sequence store (see "The adaptive code planes" above) from the unfiltered totals:
/// code_map[p][0] = code planes (| LAYOUT_IDENTITY) or one-hot rows (| LAYOUT_ONE_HOT), code_map[p][c] = the scan symbol of
/// code c (of one-hot row c - 1); escape_count[p][s] = rows of symbol s at p that get neither.  A small dynamic program over
/// the positions: the cost of a position under each of the four layouts plus RUN_COST for every change of layout between
/// neighbours (one-hot positions of 1, 2 or 3 rows are ONE layout: a run of rows).
inline void chooseLayouts(
   const std::vector<uint32_t>& totals, uint32_t n_scan, uint32_t n_bits, uint32_t positions, uint64_t row_bytes, bool allow_one_hot,
   uint64_t key_cost, std::vector<uint8_t>& code_map, std::vector<uint32_t>& escape_count
) {
   enum { TWO_PLANES = 0, THREE_PLANES = 1, IDENTITY = 2, ONE_HOT = 3, N_LAYOUTS = 4 };
   constexpr uint64_t NEVER = ~0ull >> 2;
   const uint64_t run_cost = 2 * row_bytes;
   std::vector<uint8_t> best(static_cast<size_t>(positions) * 7, 0xFF);   // the seven most frequent valid symbols, most frequent first
   std::vector<uint8_t> one_hot_rows(positions, 1);                       // rows of the position as a one-hot one
   std::vector<uint64_t> cost(static_cast<size_t>(positions) * N_LAYOUTS);
   for (uint32_t p = 0; p < positions; ++p) {
      const uint32_t* count = totals.data() + static_cast<size_t>(p) * n_scan;
      uint64_t total = 0;
      for (uint32_t symbol = 0; symbol < n_scan; ++symbol) {
         total += count[symbol];
      }
      uint32_t taken = 0;
      uint64_t carried[8] = {0, 0, 0, 0, 0, 0, 0, 0};  // rows carried by the k most frequent
      for (int k = 0; k < 7; ++k) {
         uint32_t pick = 0xFFu;
         uint32_t pick_count = 0;
         for (uint32_t symbol = 0; symbol < n_scan; ++symbol) {  // ties keep the lower symbol index in front
            if (((taken >> symbol) & 1u) == 0 && count[symbol] > pick_count) {
               pick = symbol;
               pick_count = count[symbol];
            }
         }
         best[static_cast<size_t>(p) * 7 + k] = static_cast<uint8_t>(pick);
         carried[k + 1] = carried[k] + pick_count;
         if (pick != 0xFFu) {
            taken |= 1u << pick;
         }
      }
      uint64_t* position_cost = cost.data() + static_cast<size_t>(p) * N_LAYOUTS;
      position_cost[TWO_PLANES] = 2 * row_bytes + key_cost * (total - carried[3]);
      // three mapped planes only pay where the identity layout has more (amino acids)
      position_cost[THREE_PLANES] = n_bits > 3 ? 3 * row_bytes + key_cost * (total - carried[7]) : NEVER;
      // the scan of 5 identity planes decodes 22 symbols per word and runs VALU-bound at ~0.87 of the rate of the mapped layouts
      position_cost[IDENTITY] = n_bits > 3 ? n_bits * row_bytes * 115 / 100 : n_bits * row_bytes;
      // k rows, one per symbol: one row where one symbol has (nearly) all rows — most positions of a real alignment
      position_cost[ONE_HOT] = NEVER;
      for (uint32_t k = 1; allow_one_hot && k <= 3; ++k) {
         const uint64_t with_k = k * row_bytes + key_cost * (total - carried[k]);
         if (with_k < position_cost[ONE_HOT]) {
            position_cost[ONE_HOT] = with_k;
            one_hot_rows[p] = static_cast<uint8_t>(k);
         }
      }
   }
   std::vector<uint64_t> reach(static_cast<size_t>(positions) * N_LAYOUTS);  // cheapest way to encode positions [0, p] with p in that layout
   std::vector<uint8_t> from(static_cast<size_t>(positions) * N_LAYOUTS);
   for (uint32_t p = 0; p < positions; ++p) {
      for (int layout = 0; layout < N_LAYOUTS; ++layout) {
         uint64_t before = 0;
         uint8_t previous = static_cast<uint8_t>(layout);
         if (p > 0) {
            before = NEVER;
            for (int other = 0; other < N_LAYOUTS; ++other) {
               const uint64_t candidate = reach[static_cast<size_t>(p - 1) * N_LAYOUTS + other] + (other == layout ? 0 : run_cost);
               if (candidate < before) {
                  before = candidate;
                  previous = static_cast<uint8_t>(other);
               }
            }
         }
         reach[static_cast<size_t>(p) * N_LAYOUTS + layout] = std::min(NEVER, before + cost[static_cast<size_t>(p) * N_LAYOUTS + layout]);
         from[static_cast<size_t>(p) * N_LAYOUTS + layout] = previous;
      }
   }
   code_map.assign(static_cast<size_t>(positions) * CODE_MAP_STRIDE, 0xFF);
   escape_count.assign(static_cast<size_t>(positions) * n_scan, 0);
   int layout = 0;
   for (int other = 1; other < N_LAYOUTS && positions > 0; ++other) {
      if (reach[static_cast<size_t>(positions - 1) * N_LAYOUTS + other] < reach[static_cast<size_t>(positions - 1) * N_LAYOUTS + layout]) {
         layout = other;
      }
   }
   for (uint32_t p = positions; p-- > 0;) {
      uint8_t* map = code_map.data() + static_cast<size_t>(p) * CODE_MAP_STRIDE;
      if (layout == IDENTITY) {
         map[0] = static_cast<uint8_t>(n_bits | LAYOUT_IDENTITY);
         for (uint32_t code = 1; code < CODE_MAP_STRIDE; ++code) {
            map[code] = static_cast<uint8_t>(code - 1 < n_scan && code < (1u << n_bits) ? code - 1 : 0xFFu);
         }
      } else {
         const uint32_t coded = layout == ONE_HOT ? one_hot_rows[p] : (layout == TWO_PLANES ? 3 : 7);
         map[0] = static_cast<uint8_t>(layout == ONE_HOT ? (one_hot_rows[p] | LAYOUT_ONE_HOT) : (layout == TWO_PLANES ? 2 : 3));
         uint32_t coded_mask = 0;
         for (uint32_t code = 1; code <= coded; ++code) {
            map[code] = best[static_cast<size_t>(p) * 7 + code - 1];
            if (map[code] != 0xFFu) {
               coded_mask |= 1u << map[code];
            }
         }
         for (uint32_t symbol = 0; symbol < n_scan; ++symbol) {
            if (((coded_mask >> symbol) & 1u) == 0) {
               escape_count[static_cast<size_t>(p) * n_scan + symbol] = totals[static_cast<size_t>(p) * n_scan + symbol];
            }
         }
      }
      layout = from[static_cast<size_t>(p) * N_LAYOUTS + layout];
   }
}

}  // namespace silo_gpu_layout
