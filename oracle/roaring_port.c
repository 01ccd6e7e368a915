/*
 * roaring_port.c — CPU port of the reference's Mutations scan over roaring-format containers.
 * TEST / BASELINE INFRASTRUCTURE ONLY: linked by nothing in the product; only tests/ and bench.py's
 * cpu_baseline leg load it (through oracle/cpu_port.py).
 *
 * What is restated (file:line under the reference tree, pflanze/LAPIS-SILO @ 2025-01-17):
 *   actions/mutations.cpp:64-96    addPositionToMutationCountsForMixedBitmaps  -> port_scan_mixed
 *   actions/mutations.cpp:98-136   addPositionToMutationCountsForFullBitmaps   -> port_scan_full
 *   actions/mutations.cpp:139-164  calculateMutationsPerPosition: parallel over positions, grain 300
 *   storage/sequence_store.cpp:100-211  fillIndexes / fillNBitmaps / optimizeBitmaps -> port_store_build
 *   storage/position.cpp:42-68,102-127  most numerous symbol is deleted (strict >, SYMBOLS order)
 * The bitmap type stands in for CRoaring 1.0.0 (conanfile.py:16; not vendored in the reference tree):
 * per 2^16-id chunk an array (<= 4096 sorted uint16), a 1024-word bitset or, after runOptimize, a run
 * container — the published roaring format — so the cost model follows the sparse CPU path, not the
 * dense GPU layout.  Results are checked against oracle/silo_oracle.py and the HIP kernels in tests/.
 */
#include <omp.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

enum { C_ARRAY = 1, C_BITSET = 2, C_RUN = 3 };

typedef struct {
   uint16_t key;
   uint8_t type;
   uint32_t card;
   uint32_t n; /* array: elements, run: runs, bitset: 1024 */
   union {
      uint16_t* array;
      uint64_t* bits;
      uint16_t* runs; /* pairs (start, length-1) */
   } u;
} cont_t;

typedef struct {
   uint32_t n;
   cont_t* c;
   uint64_t card;
} rbm_t;

static void rbm_free(rbm_t* b) {
   if (b == NULL) {
      return;
   }
   for (uint32_t i = 0; i < b->n; ++i) {
      free(b->c[i].u.array);
   }
   free(b->c);
   free(b);
}

/* Builds a bitmap from dense words [0, n_bits); run_optimize mirrors Roaring::runOptimize(). */
static rbm_t* rbm_from_words(const uint64_t* words, uint64_t n_bits, int run_optimize) {
   const uint64_t n_words = (n_bits + 63) / 64;
   const uint32_t n_chunks = (uint32_t)((n_bits + 65535) / 65536);
   rbm_t* out = (rbm_t*)calloc(1, sizeof(rbm_t));
   out->c = (cont_t*)calloc(n_chunks ? n_chunks : 1, sizeof(cont_t));
   for (uint32_t k = 0; k < n_chunks; ++k) {
      const uint64_t w0 = (uint64_t)k * 1024;
      const uint64_t w1 = w0 + 1024 < n_words ? w0 + 1024 : n_words;
      uint32_t card = 0, n_runs = 0;
      uint64_t carry = 0; /* previous bit */
      for (uint64_t w = w0; w < w1; ++w) {
         const uint64_t v = words[w];
         card += (uint32_t)__builtin_popcountll(v);
         n_runs += (uint32_t)__builtin_popcountll(v & ~((v << 1) | carry));
         carry = v >> 63;
      }
      if (card == 0) {
         continue;
      }
      cont_t* c = &out->c[out->n++];
      c->key = (uint16_t)k;
      c->card = card;
      const uint32_t size_run = 2 + 4 * n_runs;
      const uint32_t size_other = card <= 4096 ? 2 * card : 8192;
      if (run_optimize && size_run < size_other) {
         c->type = C_RUN;
         c->n = n_runs;
         c->u.runs = (uint16_t*)malloc((size_t)n_runs * 4);
         uint32_t r = 0;
         int in_run = 0;
         uint32_t start = 0;
         for (uint64_t w = w0; w < w1; ++w) {
            uint64_t v = words[w];
            const uint32_t base = (uint32_t)(w - w0) * 64;
            if (!in_run && v == 0) {
               continue;
            }
            for (uint32_t b = 0; b < 64; ++b) {
               const int bit = (int)((v >> b) & 1);
               if (bit && !in_run) {
                  in_run = 1;
                  start = base + b;
               } else if (!bit && in_run) {
                  in_run = 0;
                  c->u.runs[2 * r] = (uint16_t)start;
                  c->u.runs[2 * r + 1] = (uint16_t)(base + b - 1 - start);
                  ++r;
               }
            }
         }
         if (in_run) {
            const uint32_t end = (uint32_t)(w1 - w0) * 64;
            c->u.runs[2 * r] = (uint16_t)start;
            c->u.runs[2 * r + 1] = (uint16_t)(end - 1 - start);
            ++r;
         }
      } else if (card <= 4096) {
         c->type = C_ARRAY;
         c->n = card;
         c->u.array = (uint16_t*)malloc((size_t)card * 2);
         uint32_t i = 0;
         for (uint64_t w = w0; w < w1; ++w) {
            uint64_t v = words[w];
            const uint32_t base = (uint32_t)(w - w0) * 64;
            while (v) {
               c->u.array[i++] = (uint16_t)(base + (uint32_t)__builtin_ctzll(v));
               v &= v - 1;
            }
         }
      } else {
         c->type = C_BITSET;
         c->n = 1024;
         c->u.bits = (uint64_t*)calloc(1024, 8);
         memcpy(c->u.bits, words + w0, (size_t)(w1 - w0) * 8);
      }
      out->card += card;
   }
   return out;
}

static inline uint32_t bitset_range_card(const uint64_t* bits, uint32_t start, uint32_t last) { /* inclusive */
   const uint32_t w0 = start >> 6, w1 = last >> 6;
   const uint64_t m0 = ~0ull << (start & 63);
   const uint64_t m1 = ~0ull >> (63 - (last & 63));
   if (w0 == w1) {
      return (uint32_t)__builtin_popcountll(bits[w0] & m0 & m1);
   }
   uint32_t total = (uint32_t)__builtin_popcountll(bits[w0] & m0) + (uint32_t)__builtin_popcountll(bits[w1] & m1);
   for (uint32_t w = w0 + 1; w < w1; ++w) {
      total += (uint32_t)__builtin_popcountll(bits[w]);
   }
   return total;
}

static uint32_t and_card_array_array(const uint16_t* a, uint32_t na, const uint16_t* b, uint32_t nb) {
   if (na > nb) {
      const uint16_t* t = a;
      a = b;
      b = t;
      const uint32_t tn = na;
      na = nb;
      nb = tn;
   }
   uint32_t count = 0;
   if (nb > 64u * na) { /* galloping: binary search each element of the small side */
      uint32_t lo = 0;
      for (uint32_t i = 0; i < na && lo < nb; ++i) {
         uint32_t l = lo, h = nb;
         while (l < h) {
            const uint32_t m = (l + h) >> 1;
            if (b[m] < a[i]) {
               l = m + 1;
            } else {
               h = m;
            }
         }
         if (l < nb && b[l] == a[i]) {
            ++count;
         }
         lo = l;
      }
      return count;
   }
   uint32_t i = 0, j = 0;
   while (i < na && j < nb) {
      if (a[i] < b[j]) {
         ++i;
      } else if (a[i] > b[j]) {
         ++j;
      } else {
         ++count;
         ++i;
         ++j;
      }
   }
   return count;
}

static uint32_t and_card_array_bitset(const uint16_t* a, uint32_t na, const uint64_t* bits) {
   uint32_t count = 0;
   for (uint32_t i = 0; i < na; ++i) {
      count += (uint32_t)((bits[a[i] >> 6] >> (a[i] & 63)) & 1);
   }
   return count;
}

static uint32_t and_card_bitset_bitset(const uint64_t* a, const uint64_t* b) {
   uint32_t count = 0;
   for (uint32_t w = 0; w < 1024; ++w) {
      count += (uint32_t)__builtin_popcountll(a[w] & b[w]);
   }
   return count;
}

static uint32_t and_card_run_bitset(const uint16_t* runs, uint32_t nr, const uint64_t* bits) {
   uint32_t count = 0;
   for (uint32_t r = 0; r < nr; ++r) {
      count += bitset_range_card(bits, runs[2 * r], (uint32_t)runs[2 * r] + runs[2 * r + 1]);
   }
   return count;
}

static uint32_t and_card_run_array(const uint16_t* runs, uint32_t nr, const uint16_t* a, uint32_t na) {
   uint32_t count = 0, r = 0;
   for (uint32_t i = 0; i < na && r < nr; ++i) {
      while (r < nr && (uint32_t)runs[2 * r] + runs[2 * r + 1] < a[i]) {
         ++r;
      }
      if (r < nr && runs[2 * r] <= a[i]) {
         ++count;
      }
   }
   return count;
}

static uint32_t and_card_run_run(const uint16_t* a, uint32_t na, const uint16_t* b, uint32_t nb) {
   uint32_t count = 0, i = 0, j = 0;
   while (i < na && j < nb) {
      const uint32_t as = a[2 * i], ae = as + a[2 * i + 1];
      const uint32_t bs = b[2 * j], be = bs + b[2 * j + 1];
      const uint32_t lo = as > bs ? as : bs;
      const uint32_t hi = ae < be ? ae : be;
      if (lo <= hi) {
         count += hi - lo + 1;
      }
      if (ae < be) {
         ++i;
      } else {
         ++j;
      }
   }
   return count;
}

static uint32_t cont_and_card(const cont_t* a, const cont_t* b) {
   if (a->type > b->type) {
      const cont_t* t = a;
      a = b;
      b = t;
   }
   switch (a->type * 4 + b->type) {
      case C_ARRAY * 4 + C_ARRAY: return and_card_array_array(a->u.array, a->n, b->u.array, b->n);
      case C_ARRAY * 4 + C_BITSET: return and_card_array_bitset(a->u.array, a->n, b->u.bits);
      case C_ARRAY * 4 + C_RUN: return and_card_run_array(b->u.runs, b->n, a->u.array, a->n);
      case C_BITSET * 4 + C_BITSET: return and_card_bitset_bitset(a->u.bits, b->u.bits);
      case C_BITSET * 4 + C_RUN: return and_card_run_bitset(b->u.runs, b->n, a->u.bits);
      case C_RUN * 4 + C_RUN: return and_card_run_run(a->u.runs, a->n, b->u.runs, b->n);
      default: return 0;
   }
}

/* roaring_bitmap_and_cardinality */
static uint64_t rbm_and_cardinality(const rbm_t* a, const rbm_t* b) {
   uint64_t total = 0;
   uint32_t i = 0, j = 0;
   while (i < a->n && j < b->n) {
      if (a->c[i].key < b->c[j].key) {
         ++i;
      } else if (a->c[i].key > b->c[j].key) {
         ++j;
      } else {
         total += cont_and_card(&a->c[i], &b->c[j]);
         ++i;
         ++j;
      }
   }
   return total;
}

/* roaring_bitmap_contains */
static int rbm_contains(const rbm_t* b, uint32_t value) {
   const uint16_t key = (uint16_t)(value >> 16), low = (uint16_t)value;
   uint32_t l = 0, h = b->n;
   while (l < h) {
      const uint32_t m = (l + h) >> 1;
      if (b->c[m].key < key) {
         l = m + 1;
      } else {
         h = m;
      }
   }
   if (l >= b->n || b->c[l].key != key) {
      return 0;
   }
   const cont_t* c = &b->c[l];
   if (c->type == C_BITSET) {
      return (int)((c->u.bits[low >> 6] >> (low & 63)) & 1);
   }
   if (c->type == C_ARRAY) {
      uint32_t lo = 0, hi = c->n;
      while (lo < hi) {
         const uint32_t m = (lo + hi) >> 1;
         if (c->u.array[m] < low) {
            lo = m + 1;
         } else {
            hi = m;
         }
      }
      return lo < c->n && c->u.array[lo] == low;
   }
   uint32_t lo = 0, hi = c->n; /* runs: last run with start <= low */
   while (lo < hi) {
      const uint32_t m = (lo + hi) >> 1;
      if (c->u.runs[2 * m] <= low) {
         lo = m + 1;
      } else {
         hi = m;
      }
   }
   return lo > 0 && (uint32_t)c->u.runs[2 * (lo - 1)] + c->u.runs[2 * (lo - 1) + 1] >= low;
}

/* ------------------------------------------------------------------------------------------------ */
/* synthetic model: C twin of the symbol function (oracle/synth.py; k_generate_synthetic) */
typedef struct {
   uint64_t seed;
   uint32_t n_sequences, positions, n_lineages;
   const uint16_t* lineage;
   const uint32_t *lead, *trail, *mstart, *mlen;
   const uint8_t* lineage_symbol; /* [positions][n_lineages] */
   const uint8_t* reference;
   uint32_t private_threshold, ambiguous_threshold;
   uint32_t is_aa;
} synth_t;

static inline uint64_t mix64(uint64_t z) {
   z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
   z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
   return z ^ (z >> 31);
}

static inline uint8_t synth_symbol(const synth_t* m, uint32_t i, uint32_t p) {
   if (p < m->lead[i] || p >= m->positions - m->trail[i]) {
      return 0;
   }
   if (p >= m->mstart[i] && p - m->mstart[i] < m->mlen[i]) {
      return m->is_aa ? 24 : 15;
   }
   const uint64_t h = mix64((m->seed ^ ((uint64_t)i * 0x9E3779B97F4A7C15ull)) ^ ((uint64_t)p * 0xC2B2AE3D27D4EB4Full));
   if ((h & 0xFFFFFu) < m->private_threshold) {
      return (uint8_t)(1 + ((uint32_t)((h >> 20) & 0xFFFu)) % (m->is_aa ? 20u : 4u));
   }
   if (((h >> 32) & 0xFFFFFFu) < m->ambiguous_threshold) {
      return (uint8_t)((m->is_aa ? 21u : 5u) + ((uint32_t)(h >> 56)) % (m->is_aa ? 2u : 10u));
   }
   const uint8_t ls = m->lineage_symbol[(size_t)p * m->n_lineages + m->lineage[i]];
   return ls != 0xFF ? ls : m->reference[p];
}

/* ------------------------------------------------------------------------------------------------ */
/* SequenceStorePartition over a contiguous range of positions                                       */
typedef struct {
   rbm_t* bitmaps[25];
   int deleted; /* symbol whose bitmap is deleted, -1 if none */
} pos_t;

typedef struct {
   uint32_t n_sequences, pos_begin, n_positions, n_symbols, missing_symbol;
   const uint8_t* symbols_order; /* SYMBOLS iteration order */
   pos_t* positions;
   rbm_t** missing_rows; /* [n_sequences]: positions (absolute) where the row has the missing symbol */
} port_store_t;

static const uint8_t NUC_ORDER[16] = {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15};
static const uint8_t AA_ORDER[25] = {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16, 17, 18, 19, 20, 21, 22, 24, 23};

void port_store_free(port_store_t* s) {
   if (s == NULL) {
      return;
   }
   for (uint32_t p = 0; p < s->n_positions; ++p) {
      for (uint32_t k = 0; k < s->n_symbols; ++k) {
         rbm_free(s->positions[p].bitmaps[k]);
      }
   }
   for (uint32_t i = 0; i < s->n_sequences; ++i) {
      rbm_free(s->missing_rows[i]);
   }
   free(s->positions);
   free(s->missing_rows);
   free(s);
}

/* Builds the index for positions [pos_begin, pos_begin + n_positions) either from the synthetic model
 * (symbols == NULL) or from a row-major symbol matrix [n_sequences][n_positions] (model == NULL). */
port_store_t* port_store_build(const synth_t* model, const uint8_t* symbols, uint32_t n_sequences, uint32_t pos_begin, uint32_t n_positions, uint32_t is_aa) {
   port_store_t* s = (port_store_t*)calloc(1, sizeof(port_store_t));
   s->n_sequences = n_sequences;
   s->pos_begin = pos_begin;
   s->n_positions = n_positions;
   s->n_symbols = is_aa ? 25 : 16;
   s->missing_symbol = is_aa ? 24 : 15;
   s->symbols_order = is_aa ? AA_ORDER : NUC_ORDER;
   s->positions = (pos_t*)calloc(n_positions ? n_positions : 1, sizeof(pos_t));
   s->missing_rows = (rbm_t**)calloc(n_sequences ? n_sequences : 1, sizeof(rbm_t*));
   const uint64_t n_words = ((uint64_t)n_sequences + 63) / 64;
   const uint32_t n_symbols = s->n_symbols;

#pragma omp parallel
   {
      uint64_t* planes = (uint64_t*)malloc((size_t)n_symbols * (n_words ? n_words : 1) * 8);
#pragma omp for schedule(dynamic, 8)
      for (uint32_t p = 0; p < n_positions; ++p) {
         memset(planes, 0, (size_t)n_symbols * n_words * 8);
         for (uint32_t i = 0; i < n_sequences; ++i) { /* fillIndexes: ids per symbol per position */
            const uint8_t sym = model ? synth_symbol(model, i, pos_begin + p) : symbols[(size_t)i * n_positions + p];
            planes[(size_t)sym * n_words + (i >> 6)] |= 1ull << (i & 63);
         }
         pos_t* position = &s->positions[p];
         uint64_t max_count = 0;
         int max_symbol = -1;
         uint64_t cards[25];
         for (uint32_t k = 0; k < n_symbols; ++k) { /* position.cpp:42-68: strict >, SYMBOLS order */
            const uint8_t sym = s->symbols_order[k];
            uint64_t card = 0;
            if (sym != s->missing_symbol) { /* the missing symbol is never indexed */
               const uint64_t* w = planes + (size_t)sym * n_words;
               for (uint64_t x = 0; x < n_words; ++x) {
                  card += (uint64_t)__builtin_popcountll(w[x]);
               }
            }
            cards[sym] = card;
            if (card > max_count) {
               max_count = card;
               max_symbol = sym;
            }
         }
         position->deleted = max_symbol; /* deleteMostNumerousBitmap, position.cpp:102-127 */
         for (uint32_t sym = 0; sym < n_symbols; ++sym) {
            if ((int)sym == max_symbol || sym == s->missing_symbol || cards[sym] == 0) {
               position->bitmaps[sym] = rbm_from_words(planes, 0, 1); /* empty */
            } else {
               position->bitmaps[sym] = rbm_from_words(planes + (size_t)sym * n_words, n_sequences, 1);
            }
         }
      }
      free(planes);

      /* fillNBitmaps: row-wise positions of the missing symbol */
      const uint64_t row_words = ((uint64_t)pos_begin + n_positions + 63) / 64;
      uint64_t* row = (uint64_t*)malloc((row_words ? row_words : 1) * 8);
#pragma omp for schedule(dynamic, 1024)
      for (uint32_t i = 0; i < n_sequences; ++i) {
         memset(row, 0, row_words * 8);
         for (uint32_t p = 0; p < n_positions; ++p) {
            const uint8_t sym = model ? synth_symbol(model, i, pos_begin + p) : symbols[(size_t)i * n_positions + p];
            if (sym == s->missing_symbol) {
               const uint32_t q = pos_begin + p;
               row[q >> 6] |= 1ull << (q & 63);
            }
         }
         s->missing_rows[i] = rbm_from_words(row, (uint64_t)pos_begin + n_positions, 1);
      }
      free(row);
   }
   return s;
}

rbm_t* port_filter_from_words(const uint64_t* words, uint32_t n_bits) {
   return rbm_from_words(words, n_bits, 1); /* mutations.cpp:53-55 runOptimize()s mutable filters */
}
void port_filter_free(rbm_t* filter) {
   rbm_free(filter);
}
uint64_t port_filter_cardinality(const rbm_t* filter) {
   return filter->card;
}
uint64_t port_and_cardinality(const rbm_t* a, const rbm_t* b) {
   return rbm_and_cardinality(a, b);
}
int port_contains(const rbm_t* a, uint32_t value) {
   return rbm_contains(a, value);
}

/* addPositionToMutationCountsForMixedBitmaps (mutations.cpp:64-96), uint32 wrap-around arithmetic */
static void scan_mixed_position(const port_store_t* s, const rbm_t* filter, uint32_t p, uint32_t* counts /* [n_symbols] */) {
   const pos_t* position = &s->positions[p];
   const uint32_t absolute = s->pos_begin + p;
   for (uint32_t k = 0; k < s->n_symbols; ++k) {
      const uint8_t symbol = s->symbols_order[k];
      if ((int)symbol == position->deleted) {
         counts[symbol] += (uint32_t)filter->card;
         for (uint32_t ci = 0; ci < filter->n; ++ci) { /* for (idx : *filter) missing[idx].contains(pos) */
            const cont_t* c = &filter->c[ci];
            const uint32_t base = (uint32_t)c->key << 16;
            if (c->type == C_ARRAY) {
               for (uint32_t e = 0; e < c->n; ++e) {
                  counts[symbol] -= (uint32_t)rbm_contains(s->missing_rows[base + c->u.array[e]], absolute);
               }
            } else if (c->type == C_RUN) {
               for (uint32_t r = 0; r < c->n; ++r) {
                  const uint32_t first = base + c->u.runs[2 * r];
                  for (uint32_t e = 0; e <= c->u.runs[2 * r + 1]; ++e) {
                     counts[symbol] -= (uint32_t)rbm_contains(s->missing_rows[first + e], absolute);
                  }
               }
            } else {
               for (uint32_t w = 0; w < 1024; ++w) {
                  uint64_t v = c->u.bits[w];
                  while (v) {
                     const uint32_t idx = base + w * 64 + (uint32_t)__builtin_ctzll(v);
                     counts[symbol] -= (uint32_t)rbm_contains(s->missing_rows[idx], absolute);
                     v &= v - 1;
                  }
               }
            }
         }
         continue;
      }
      const uint32_t symbol_count = (uint32_t)rbm_and_cardinality(filter, position->bitmaps[symbol]);
      counts[symbol] += symbol_count;
      if (position->deleted >= 0) {
         counts[position->deleted] -= symbol_count;
      }
   }
}

/* addPositionToMutationCountsForFullBitmaps (mutations.cpp:98-136) */
static void scan_full_position(const port_store_t* s, uint32_t p, uint32_t* counts) {
   const pos_t* position = &s->positions[p];
   const uint32_t absolute = s->pos_begin + p;
   for (uint32_t k = 0; k < s->n_symbols; ++k) {
      const uint8_t symbol = s->symbols_order[k];
      if ((int)symbol == position->deleted) {
         counts[symbol] += s->n_sequences;
         for (uint32_t i = 0; i < s->n_sequences; ++i) {
            counts[symbol] -= (uint32_t)rbm_contains(s->missing_rows[i], absolute);
         }
         continue;
      }
      const uint32_t symbol_count = (uint32_t)position->bitmaps[symbol]->card;
      counts[symbol] += symbol_count;
      if (position->deleted >= 0) {
         counts[position->deleted] -= symbol_count;
      }
   }
}

/* calculateMutationsPerPosition (mutations.cpp:139-164): parallel over positions in chunks of `grain`
 * (the reference uses tbb::blocked_range with grain 300).  filter == NULL -> the full-bitmap path.
 * counts_out is [n_positions][n_symbols], ACCUMULATED into.  Returns the elapsed seconds. */
double port_mutations_scan(const port_store_t* s, const rbm_t* filter, uint32_t* counts_out, int n_threads, int grain) {
   if (n_threads > 0) {
      omp_set_num_threads(n_threads);
   }
   if (grain <= 0) {
      grain = 300;
   }
   const double t0 = omp_get_wtime();
   const int64_t n_chunks = ((int64_t)s->n_positions + grain - 1) / grain;
#pragma omp parallel for schedule(dynamic, 1)
   for (int64_t chunk = 0; chunk < n_chunks; ++chunk) {
      const uint32_t begin = (uint32_t)(chunk * grain);
      const uint32_t end = begin + (uint32_t)grain < s->n_positions ? begin + (uint32_t)grain : s->n_positions;
      for (uint32_t p = begin; p < end; ++p) {
         if (filter != NULL) {
            scan_mixed_position(s, filter, p, counts_out + (size_t)p * s->n_symbols);
         } else {
            scan_full_position(s, p, counts_out + (size_t)p * s->n_symbols);
         }
      }
   }
   return omp_get_wtime() - t0;
}

int port_max_threads(void) {
   return omp_get_max_threads();
}

/* container census, for the cost-model note in DESIGN.md */
void port_store_census(const port_store_t* s, uint64_t out[4]) { /* arrays, bitsets, runs, bytes */
   memset(out, 0, 4 * sizeof(uint64_t));
   for (uint32_t p = 0; p < s->n_positions; ++p) {
      for (uint32_t k = 0; k < s->n_symbols; ++k) {
         const rbm_t* b = s->positions[p].bitmaps[k];
         for (uint32_t i = 0; b != NULL && i < b->n; ++i) {
            const cont_t* c = &b->c[i];
            out[c->type - 1] += 1;
            out[3] += c->type == C_ARRAY ? 2ull * c->n : c->type == C_RUN ? 4ull * c->n : 8192ull;
         }
      }
   }
}
