// layout_choice.h — the layout of every position of a finalized sequence store (host code, shared by silo_gpu.hip and the
// CPU unit tests: no device is involved).  See "The adaptive code planes" in silo_gpu.hip and DESIGN.md §2.
#pragma once
#include <stdint.h>

#include <algorithm>
#include <vector>

namespace silo_gpu_layout {

// code_map[p][0]: the number of plane rows of the position, with
constexpr uint8_t LAYOUT_IDENTITY = 0x80;  // identity code planes (code = index of the valid mutation symbol + 1, no escapes), or
constexpr uint8_t LAYOUT_ONE_HOT = 0x40;   // ONE-HOT rows: k = 0..3 rows, row j = the rows of the symbol code_map[p][1 + j]
// ... of which the position's MOST NUMEROUS valid symbol (code_map[p][IMPLICIT_SLOT]) has no row and no keys at all: every row has
// exactly one symbol at a position, so under a filter F its count is |F| - (rows of F with no valid symbol there: the runs of
// the missing symbol, the ambiguity codes) - (the counts of the other valid symbols) — what the reference does with the bitmap it
// deletes (position.cpp:102-127, mutations.cpp:74-95).  The k rows are those of the NEXT most numerous symbols.
constexpr uint8_t LAYOUT_IMPLICIT = 0x20;
constexpr uint8_t LAYOUT_ROWS_MASK = 0x0F;  // the plane rows of the position
constexpr uint32_t CODE_MAP_STRIDE = 8;    // bytes of code_map per position: [0] = layout, [c] = scan symbol of code c (1..7), 0xFF = unused
constexpr uint32_t IMPLICIT_SLOT = 7;      // code_map[p][7] of a LAYOUT_IMPLICIT position: the symbol that is derived (one-hot rows use 1..3)
// what an escape key costs a scan, in plane bytes: the escape pass takes its 4-byte keys at ~0.4 of the rate at which the row
// kernel streams planes (60 M keys in 81 us beside 6.4 TB/s: ~9 plane bytes per key); the optimum is flat between 6 and 12
// (profiles/r03_notes.md; round 2's 8-byte keys: 16, flat between 10 and 24)
constexpr uint32_t KEY_COST_BYTES = 10;
// what a further kind of plane-scan launch costs a scan (pipeline ramp, tail, the launch boundary: 20-40 us), in plane bytes
constexpr uint64_t LAUNCH_COST_BYTES = 192ull << 20;
// one-hot rows: not at all / a row for every stored symbol / the most numerous symbol derived (LAYOUT_IMPLICIT)
enum OneHotMode : int { ONE_HOT_OFF = 0, ONE_HOT_ROWS = 1, ONE_HOT_IMPLICIT = 2 };

/// The layout of every position of a sequence store (see "The adaptive code planes" above) from the unfiltered totals:
/// code_map[p][0] = code planes (| LAYOUT_IDENTITY) or one-hot rows (| LAYOUT_ONE_HOT, | LAYOUT_IMPLICIT), code_map[p][c] = the
/// scan symbol of code c (of one-hot row c - 1); escape_count[p][s] = rows of symbol s at p that get neither.  A small dynamic
/// program over the positions: the cost of a position under each of the four layouts plus RUN_COST for every change of layout
/// between neighbours (one-hot positions of any number of rows are ONE layout: a run of rows) — run once for every subset of
/// the three code-plane layouts, each of which costs the scan a launch of its own (LAUNCH_COST_BYTES): the cheapest subset wins.
inline void chooseLayouts(
   const std::vector<uint32_t>& totals, uint32_t n_scan, uint32_t n_bits, uint32_t positions, uint64_t row_bytes, int one_hot_mode,
   uint64_t key_cost, std::vector<uint8_t>& code_map, std::vector<uint32_t>& escape_count, uint64_t launch_cost = LAUNCH_COST_BYTES
) {
   enum { TWO_PLANES = 0, THREE_PLANES = 1, IDENTITY = 2, ONE_HOT = 3, N_LAYOUTS = 4 };
   constexpr uint64_t NEVER = ~0ull >> 2;
   const bool allow_one_hot = one_hot_mode != ONE_HOT_OFF;
   const bool implicit = one_hot_mode == ONE_HOT_IMPLICIT;
   const uint64_t run_cost = 2 * row_bytes;
   std::vector<uint8_t> best(static_cast<size_t>(positions) * 7, 0xFF);   // the seven most frequent valid symbols, most frequent first
   std::vector<uint8_t> one_hot_rows(positions, implicit ? 0 : 1);        // rows of the position as a one-hot one
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
      // k rows, one per symbol: one row where one symbol has (nearly) all rows — most positions of a real alignment —, none
      // where that symbol is derived (implicit: the k rows are those of the symbols behind it)
      position_cost[ONE_HOT] = NEVER;
      for (uint32_t k = implicit ? 0 : 1; allow_one_hot && k <= 3; ++k) {
         const uint64_t with_k = k * row_bytes + key_cost * (total - carried[implicit ? k + 1 : k]);
         if (with_k < position_cost[ONE_HOT]) {
            position_cost[ONE_HOT] = with_k;
            one_hot_rows[p] = static_cast<uint8_t>(k);
         }
      }
   }
   // the dynamic program, once per subset of the code-plane layouts (bit l of `allowed`: layout l may be used)
   std::vector<uint64_t> reach(static_cast<size_t>(positions) * N_LAYOUTS);  // cheapest way to encode positions [0, p] with p in that layout
   std::vector<uint8_t> from(static_cast<size_t>(positions) * N_LAYOUTS);
   std::vector<uint8_t> chosen(positions, IDENTITY), candidate(positions);
   uint64_t chosen_cost = NEVER;
   for (uint32_t allowed = 0; allowed < 8 && positions > 0; ++allowed) {
      const auto usable = [&](int layout) { return layout == ONE_HOT ? allow_one_hot : ((allowed >> layout) & 1u) != 0; };
      if ((!allow_one_hot && allowed == 0) || (usable(THREE_PLANES) && n_bits <= 3)) {
         continue;
      }
      for (uint32_t p = 0; p < positions; ++p) {
         for (int layout = 0; layout < N_LAYOUTS; ++layout) {
            uint64_t before = 0;
            uint8_t previous = static_cast<uint8_t>(layout);
            if (p > 0) {
               before = NEVER;
               for (int other = 0; other < N_LAYOUTS; ++other) {
                  const uint64_t reached = reach[static_cast<size_t>(p - 1) * N_LAYOUTS + other] + (other == layout ? 0 : run_cost);
                  if (reached < before) {
                     before = reached;
                     previous = static_cast<uint8_t>(other);
                  }
               }
            }
            const uint64_t here = usable(layout) ? cost[static_cast<size_t>(p) * N_LAYOUTS + layout] : NEVER;
            reach[static_cast<size_t>(p) * N_LAYOUTS + layout] = std::min(NEVER, before + here);
            from[static_cast<size_t>(p) * N_LAYOUTS + layout] = previous;
         }
      }
      int layout = 0;
      for (int other = 1; other < N_LAYOUTS; ++other) {
         if (reach[static_cast<size_t>(positions - 1) * N_LAYOUTS + other] < reach[static_cast<size_t>(positions - 1) * N_LAYOUTS + layout]) {
            layout = other;
         }
      }
      uint64_t subset_cost = reach[static_cast<size_t>(positions - 1) * N_LAYOUTS + layout];
      if (subset_cost >= NEVER) {
         continue;
      }
      uint32_t used = 0;
      for (uint32_t p = positions; p-- > 0;) {
         candidate[p] = static_cast<uint8_t>(layout);
         used |= 1u << layout;
         layout = from[static_cast<size_t>(p) * N_LAYOUTS + layout];
      }
      for (int code_planes = 0; code_planes < ONE_HOT; ++code_planes) {
         subset_cost += ((used >> code_planes) & 1u) != 0 ? launch_cost : 0;
      }
      if (subset_cost < chosen_cost) {
         chosen_cost = subset_cost;
         chosen = candidate;
      }
   }
   code_map.assign(static_cast<size_t>(positions) * CODE_MAP_STRIDE, 0xFF);
   escape_count.assign(static_cast<size_t>(positions) * n_scan, 0);
   for (uint32_t p = 0; p < positions; ++p) {
      const int layout = chosen[p];
      uint8_t* map = code_map.data() + static_cast<size_t>(p) * CODE_MAP_STRIDE;
      if (layout == IDENTITY) {
         map[0] = static_cast<uint8_t>(n_bits | LAYOUT_IDENTITY);
         for (uint32_t code = 1; code < CODE_MAP_STRIDE; ++code) {
            map[code] = static_cast<uint8_t>(code - 1 < n_scan && code < (1u << n_bits) ? code - 1 : 0xFFu);
         }
         continue;
      }
      const bool derived = layout == ONE_HOT && implicit;
      const uint32_t coded = layout == ONE_HOT ? one_hot_rows[p] : (layout == TWO_PLANES ? 3 : 7);
      map[0] = static_cast<uint8_t>(layout == ONE_HOT ? (one_hot_rows[p] | LAYOUT_ONE_HOT | (derived ? LAYOUT_IMPLICIT : 0)) : (layout == TWO_PLANES ? 2 : 3));
      uint32_t coded_mask = 0;
      if (derived) {  // (a position without any valid symbol derives symbol 0: |F| - the rows without a valid symbol - 0 = 0)
         map[IMPLICIT_SLOT] = best[static_cast<size_t>(p) * 7] != 0xFFu ? best[static_cast<size_t>(p) * 7] : 0;
         coded_mask |= 1u << map[IMPLICIT_SLOT];
      }
      for (uint32_t code = 1; code <= coded; ++code) {
         map[code] = best[static_cast<size_t>(p) * 7 + code - 1 + (derived ? 1 : 0)];
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
}

}  // namespace silo_gpu_layout
