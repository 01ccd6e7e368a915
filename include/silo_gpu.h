/*
 * silo_gpu.h — C ABI of the MI355X (gfx950) device side of the SILO mutation-filter hot path.
 *
 * This is the drop-in boundary described in SURVEY.md §8(b): plain pointers and sizes, no C++/torch
 * types, 0 on success / negative silo_gpu_status on failure, never throws.  The reference has no FFI
 * layer; each entry point below names the reference C++ interface (file:line under the reference
 * tree) whose arithmetic it replaces.
 *
 * Data model: a *store* is one DatabasePartition (database_partition.h:39-112) with
 * `sequence_count` rows.  Every sequence store in it (nucleotide segment or gene) is a
 * SequenceStorePartition (sequence_store.h:34-88) restated as dense bitsets in HBM:
 *   row_words  Wp = roundup(ceil(N/64), 32)   (256-byte aligned rows, tail bits zero)
 *   scan planes   [position][valid mutation symbol][Wp]   uint64, bit i of word w = sequence 64*w+i
 *   extra planes  [extra dense symbol][position][Wp]      (the missing symbol N / X, ...)
 *   sparse symbols (IUPAC ambiguity codes) as sorted (position, symbol, sequence id) triples.
 */
#ifndef SILO_GPU_H
#define SILO_GPU_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum {
   SILO_GPU_OK = 0,
   SILO_GPU_ERR_INVALID_ARGUMENT = -1,
   SILO_GPU_ERR_OUT_OF_MEMORY = -2,
   SILO_GPU_ERR_HIP = -3,         /* a HIP runtime call failed; see silo_gpu_last_error() */
   SILO_GPU_ERR_NO_DEVICE = -4,   /* no gfx950 device visible: the product path has no CPU fallback */
   SILO_GPU_ERR_PROGRAM_TOO_LARGE = -5,
   SILO_GPU_ERR_UNSUPPORTED = -6
} silo_gpu_status;

/* Alphabets.  Symbol ids are the reference enum values:
 * nucleotide_symbols.h:15-32 (GAP A C G T R Y S W K M B D H V N = 0..15),
 * aa_symbols.h:15-41 (GAP A C D E F G H I K L M N P Q R S T V W Y B Z STOP X = 0..24). */
enum { SILO_GPU_ALPHABET_NUCLEOTIDE = 0, SILO_GPU_ALPHABET_AMINO_ACID = 1 };
enum { SILO_GPU_NUC_SYMBOLS = 16, SILO_GPU_AA_SYMBOLS = 25, SILO_GPU_MAX_SYMBOLS = 25 };
#define SILO_GPU_SYMBOL_NONE 0xFFu

typedef struct silo_gpu_store silo_gpu_store; /* opaque; one per device shard */

/* One SequenceStorePartition (sequence_store.h:51-56). */
typedef struct {
   uint32_t alphabet;         /* SILO_GPU_ALPHABET_* */
   uint32_t positions;        /* reference_sequence.size() */
   const uint8_t* reference;  /* [positions] symbol ids (reference_genomes.cpp) */
   /* Dense planes to allocate.  scan_symbols must be the alphabet's VALID_MUTATION_SYMBOLS in
    * their declared order (nucleotide_symbols.h:61-67, aa_symbols.h:56-79): they form the
    * [position][symbol] block the Mutations scan streams.  extra_symbols are further dense planes
    * (normally the missing symbol, SYMBOL_MISSING).  Symbols in neither list are kept sparse. */
   uint32_t n_scan_symbols;
   const uint8_t* scan_symbols;
   uint32_t n_extra_symbols;
   const uint8_t* extra_symbols;
} silo_gpu_seqstore_desc;

typedef struct {
   int32_t device;            /* HIP device ordinal */
   uint32_t sequence_count;   /* DatabasePartition::sequence_count (capacity; rows are 0..N-1) */
   uint32_t n_seqstores;
   const silo_gpu_seqstore_desc* seqstores;
} silo_gpu_store_desc;

/* How finalize lays a store out — per STORE (two engines of one process may differ); a field left at SILO_GPU_OPTION_DEFAULT
 * follows the process-wide silo_gpu_tune knob of the same meaning (those are for probes and tests).  Set before finalize. */
#define SILO_GPU_OPTION_DEFAULT INT32_MIN
typedef struct silo_gpu_store_options {
   int32_t layout;           /* as SILO_GPU_TUNE_COMPACT_INDEX: < 0 keep the build-time identity planes, 0 re-encode every position into its
                                cheapest layout (the most numerous symbol derived where the missing symbol is kept as runs), 2 without one-hot
                                rows, 3 with a one-hot row for the most numerous symbol too */
   int32_t missing_runs;     /* as SILO_GPU_TUNE_MISSING_RUNS: < 0 keeps the plane of the missing symbol */
   int32_t key_cost;         /* as SILO_GPU_TUNE_KEY_COST: > 0 = plane bytes an escape key costs in the choice of layouts */
   int32_t launch_cost_kib;  /* as SILO_GPU_TUNE_LAUNCH_COST */
} silo_gpu_store_options;

/* ---- lifetime ------------------------------------------------------------------------------- */

/* Allocates zeroed planes in HBM.  Replaces SequenceStorePartition's constructor
 * (sequence_store.cpp:21-29).  Fails with SILO_GPU_ERR_NO_DEVICE when no GPU is present, and with SILO_GPU_ERR_UNSUPPORTED for a
 * device other than that of the process's first store: one process serves one GPU. */
int silo_gpu_store_create(const silo_gpu_store_desc* desc, silo_gpu_store** out);
void silo_gpu_store_destroy(silo_gpu_store* store);

/* NULL resets every field to SILO_GPU_OPTION_DEFAULT.  Stores that are finalized already keep their layout. */
int silo_gpu_store_set_options(silo_gpu_store* store, const silo_gpu_store_options* options);

uint32_t silo_gpu_store_sequence_count(const silo_gpu_store* store);
uint32_t silo_gpu_store_row_words(const silo_gpu_store* store); /* Wp, in uint64 words */
uint64_t silo_gpu_store_device_bytes(const silo_gpu_store* store);
/* Free and total memory of the store's device (hipMemGetInfo), for callers that choose between the one-pass and the two-pass build. */
int silo_gpu_store_memory_info(const silo_gpu_store* store, uint64_t* free_bytes, uint64_t* total_bytes);

/* ---- index build: SequenceStorePartition::fill / interpret (sequence_store.cpp:31-66,100-220) - */

/* Transposes a batch of aligned sequences into the planes on the device.
 * `chars` is host memory, row-major [n_sequences][positions], the characters of the alignment
 * (charToSymbol: nucleotide_symbols.cpp:46-85, aa_symbols.cpp:62-117).  `is_null[i] != 0` marks a
 * missing genome: the missing symbol at every position (sequence_store.cpp:166-169); its chars are
 * ignored.  is_null may be NULL.  first_sequence + n_sequences <= sequence_count.
 * An illegal character fails with SILO_GPU_ERR_INVALID_ARGUMENT (sequence_store.cpp:118-122). */
int silo_gpu_store_append_sequences(
   silo_gpu_store* store, uint32_t seqstore_id, uint32_t first_sequence, uint32_t n_sequences,
   const char* chars, const uint8_t* is_null
);

/* ---- import from the reference's own storage form (the roaring payloads of a snapshot) ---------------
 * A reference Position (position.h:27-37) is one roaring bitmap per symbol — written by saveDatabaseState in CRoaring's
 * portable serialization (roaring_serialize.h:17-45) — with the most numerous symbol stored FLIPPED (its complement) or
 * DELETED (empty; position.cpp:42-127), and the missing symbol kept ROW-wise (missing_symbol_bitmaps[row] = the positions
 * where the row has N / X, sequence_store.cpp:153-190).  silo_gpu_store_import_missing_rows first (the deleted symbol of a
 * position is "no other symbol and not missing"), then silo_gpu_store_import_position per position; finalize as usual.
 * The containers are expanded on the device.  The Boost archive framing around the payloads is the caller's to strip: it is
 * not read here (nothing in this image to pin it against), and the portable format itself is restated from its published
 * specification — parity unpinned.  SILO_GPU_SYMBOL_NONE = no flipped / deleted symbol. */
typedef struct silo_gpu_roaring_payload {
   uint32_t symbol;    /* reference enum value; ignored by silo_gpu_store_import_missing_rows */
   const void* bytes;  /* portable-format roaring bitmap, host memory */
   size_t n_bytes;
} silo_gpu_roaring_payload;
int silo_gpu_store_import_missing_rows(
   silo_gpu_store* store, uint32_t seqstore_id, uint32_t first_sequence, uint32_t n_sequences, const silo_gpu_roaring_payload* rows
);
int silo_gpu_store_import_position(
   silo_gpu_store* store, uint32_t seqstore_id, uint32_t position, const silo_gpu_roaring_payload* bitmaps, uint32_t n_bitmaps,
   uint32_t flipped_symbol, uint32_t deleted_symbol
);

/* Two-pass build of a sequence store — for stores too large to hold their build-time planes beside the finished ones (the
 * build-time planes are 3 / 5 per position; a finished nucleotide store about one): stream the sequences TWICE.
 *   silo_gpu_store_build_pass(store, id, 1)   before the first sequence: the appends / generate calls that follow only COUNT
 *                                             the valid symbols per position (no plane is allocated);
 *   silo_gpu_store_build_pass(store, id, 2)   after all sequences have been seen once: the layout of every position is chosen
 *                                             from the counts, the adaptive planes are allocated, and the same appends /
 *                                             generate calls, repeated, write every row straight into them;
 *   silo_gpu_store_finalize / _finalize_seqstore as usual (the escape keys are sorted; the missing symbol becomes runs).
 * A store that would keep its identity planes anyway (short rows, compact_scan_index off) simply builds them in the second
 * pass.  silo_gpu_store_build_mode: 0 ordinary, 1 counting, 2 encoding.  The roaring import is not available in this mode. */
int silo_gpu_store_build_pass(silo_gpu_store* store, uint32_t seqstore_id, int pass);
int silo_gpu_store_build_mode(const silo_gpu_store* store, uint32_t seqstore_id);

/* Sorts the sparse triples gathered by append/generate; call once after the last append and
 * before any query.  (The reference's optimizeBitmaps, sequence_store.cpp:192-211, has no dense
 * analogue: flipped/deleted bitmaps are storage tricks) */
int silo_gpu_store_finalize(silo_gpu_store* store);

/* Synthetic SARS-CoV-2-shaped data written straight into the planes (bench / parity tests only;
 * the model is specified in DESIGN.md §6 and restated on the CPU in oracle/synth.py).
 * All arrays are host memory.  lineage_symbol is [positions][n_lineages], 0xFF = "reference". */
typedef struct {
   uint64_t seed;
   uint32_t n_lineages;
   const uint16_t* lineage_of_sequence; /* [N] */
   const uint32_t* lead_gap;            /* [N] positions [0, lead_gap) are GAP */
   const uint32_t* trail_gap;           /* [N] positions [P - trail_gap, P) are GAP */
   const uint32_t* missing_start;       /* [N] run of the missing symbol */
   const uint32_t* missing_len;         /* [N] */
   const uint8_t* lineage_symbol;       /* [positions][n_lineages] */
   uint32_t private_threshold;          /* of 2^20: P(private substitution) per cell */
   uint32_t ambiguous_threshold;        /* of 2^24: P(sparse ambiguity code) per cell */
   /* Position-range shards: the store holds positions [position_offset, position_offset + positions)
    * of a genome of total_positions (0 = the store's own length); lineage_symbol covers the slice. */
   uint32_t position_offset;
   uint32_t total_positions;
} silo_gpu_synth_desc;
int silo_gpu_store_generate_synthetic(
   silo_gpu_store* store, uint32_t seqstore_id, const silo_gpu_synth_desc* synth
);

/* ---- device buffers owned by the caller -------------------------------------------------------- */

/* Row-sized (Wp words) bitset in HBM, zero-initialised.  Used for filter results and for host-built
 * metadata bitsets (pango lineage sets: pango_lineage_column.cpp:57-77, uploaded once). */
int silo_gpu_bitset_alloc(const silo_gpu_store* store, uint64_t** out_dev);
int silo_gpu_bitset_upload(const silo_gpu_store* store, uint64_t* dst_dev, const uint64_t* src_host, size_t n_words, void* stream);
int silo_gpu_bitset_download(const silo_gpu_store* store, uint64_t* dst_host, const uint64_t* src_dev, size_t n_words, void* stream);
/* bit i = membership_by_lineage[lineage_of_sequence[i]] for a store filled by generate_synthetic. */
int silo_gpu_bitset_from_lineages(const silo_gpu_store* store, uint64_t* dst_dev, const uint8_t* membership_by_lineage, uint32_t n_lineages, void* stream);
/* Dictionary-encoded metadata column kept on the device: bit i = membership_by_value[value_ids_dev[i]].
 * This is how a metadata predicate (pango lineage, pango_lineage_filter.cpp:37-59) enters the device path. */
int silo_gpu_upload_u32(const uint32_t* src_host, size_t n, uint32_t** out_dev);
int silo_gpu_bitset_from_value_ids(const silo_gpu_store* store, uint64_t* dst_dev, const uint32_t* value_ids_dev, const uint8_t* membership_by_value, uint32_t n_values, void* stream);
void silo_gpu_free(void* dev_ptr);
int silo_gpu_malloc(size_t bytes, void** out_dev);
int silo_gpu_memcpy_d2h(void* dst_host, const void* src_dev, size_t bytes, void* stream); /* synchronises */
int silo_gpu_memcpy_h2d(void* dst_dev, const void* src_host, size_t bytes, void* stream); /* synchronises */
/* Result tables fetched without stalling the stream: page-locked host memory + a copy that is only enqueued
 * (wait for it with an event recorded after it, silo_gpu_event_synchronize).  The host mirror fetches the
 * counts[P][S] table of sequence store k while the scans of stores k+1.. are still running. */
int silo_gpu_host_alloc(size_t bytes, void** out_host);
void silo_gpu_host_free(void* host);
int silo_gpu_memcpy_d2h_async(void* dst_pinned_host, const void* src_dev, size_t bytes, void* stream);
int silo_gpu_stream_synchronize(void* stream);
/* A non-blocking HIP stream (does not synchronise with the null stream); every `void* stream` parameter of
 * this ABI accepts one, or NULL for the null stream. */
int silo_gpu_stream_create(void** out_stream);
/* Selects the HIP device of the CALLING host thread (HIP keeps the current device per thread, starting at 0): a thread that
 * creates streams, events or buffers for a store on device r selects r first.  Entry points that take a store select its
 * device themselves. */
int silo_gpu_set_device(int device);
void silo_gpu_stream_destroy(void* stream);

/* Device pointer of the one-hot plane of (seqstore, position, symbol) when the store holds one: the extra symbols
 * (the missing symbol N / X).  NULL for the valid mutation symbols — they live in the bit-sliced scan planes
 * ([position][code bit][Wp], code = index among the scan symbols + 1; 3 planes for 5 nucleotide symbols, 5 for 22
 * amino-acid symbols) — and for sparsely stored symbols; silo_gpu_store_sparse_plane materialises either.
 * Replaces SequenceStorePartition::getBitmap (sequence_store.cpp:92-98); positions are 0-based as there. */
const uint64_t* silo_gpu_store_plane(const silo_gpu_store* store, uint32_t seqstore_id, uint32_t position, uint32_t symbol);

/* Materialises the one-hot bitset of any symbol at a position into dst_dev (Wp words, overwritten): decoded from the
 * code planes (valid mutation symbols), copied (extra symbols) or scattered from the sorted keys (sparse symbols). */
int silo_gpu_store_sparse_plane(const silo_gpu_store* store, uint32_t seqstore_id, uint32_t position, uint32_t symbol, uint64_t* dst_dev, void* stream);

/* ---- K3: fused filter-expression evaluator --------------------------------------------------------
 * Replaces Operator::evaluate() of IndexScan / Complement / Intersection / Union / Threshold /
 * Full / Empty / BitmapSelection (operators/index_scan.cpp:28-30, complement.cpp:50-54,
 * intersection.cpp:62-127, union.cpp:35-45, threshold.cpp:64-138, full.cpp:24-28, empty.cpp,
 * bitmap_selection.cpp:33-52): the whole operator tree of one partition is one launch.
 *
 * A bit-program is a register machine over 64-bit slots, one program run per bitset word.
 * Instruction word 0: op | dst<<8 | a<<16 | b<<24, word 1: imm.   Slot 0 holds the result. */
enum {
   SILO_GPU_OP_LOAD = 0,     /* dst = leaves[imm][w]                                  (IndexScan) */
   SILO_GPU_OP_ZERO = 1,     /* dst = 0                                               (Empty)     */
   SILO_GPU_OP_ONES = 2,     /* dst = valid(w): all rows < sequence_count             (Full)      */
   SILO_GPU_OP_NOT = 3,      /* dst = ~a & valid(w)                                   (Complement)*/
   SILO_GPU_OP_AND = 4,      /* dst = a & b                                           (Intersection) */
   SILO_GPU_OP_OR = 5,       /* dst = a | b                                           (Union)     */
   SILO_GPU_OP_ANDNOT = 6,   /* dst = a & ~b                                          (Intersection, negated child) */
   SILO_GPU_OP_CNT_ADD = 7,  /* bit-sliced counter in slots dst..dst+b-1 += slot a    (Threshold) */
   SILO_GPU_OP_CNT_GE = 8,   /* dst = (counter in slots a..a+b-1) >= imm                           */
   SILO_GPU_OP_CNT_EQ = 9,   /* dst = (counter in slots a..a+b-1) == imm                           */
   SILO_GPU_OP_MOV = 10,     /* dst = a */
   /* n-ary forms over a run of consecutive leaves, imm = first_leaf | count << 16 (count >= 1): the leaves
    * are fetched 8 at a time with independent loads (flat Or / And / N-Of over stored columns). */
   SILO_GPU_OP_OR_N = 11,          /* dst = leaf[first] | ... | leaf[first+count-1]          (Union)        */
   SILO_GPU_OP_AND_N = 12,         /* dst = leaf[first] & ... & leaf[first+count-1]          (Intersection) */
   SILO_GPU_OP_CNT_ADD_N = 13,     /* counter dst..dst+b-1 += each leaf of the run           (Threshold)    */
   SILO_GPU_OP_CNT_ADD_NOT_N = 14  /* counter dst..dst+b-1 += each (~leaf & valid) of the run (negated children) */
};
enum { SILO_GPU_MAX_INSTRUCTIONS = 320, SILO_GPU_MAX_LEAVES = 128, SILO_GPU_MAX_SLOTS = 32 };
/* A source operand (a, b) >= SILO_GPU_LEAF_OPERAND reads leaf (operand - SILO_GPU_LEAF_OPERAND) directly, so a
 * leaf needs no LOAD instruction of its own; destinations are always slots. */
enum { SILO_GPU_LEAF_OPERAND = 32 };
/* Cardinalities are accumulated into SILO_GPU_COUNT_SHARDS consecutive uint64 counters (to keep device
 * atomics off a single word); the total is their sum.  Buffers passed as out_count_dev have that many. */
enum { SILO_GPU_COUNT_SHARDS = 64 };

typedef struct {
   uint32_t n_instructions;
   const uint32_t* code;            /* 2 * n_instructions words, host memory */
   uint32_t n_leaves;
   const uint64_t* const* leaves;   /* device pointers, each Wp words, host array */
   uint32_t n_slots;                /* slots used, <= SILO_GPU_MAX_SLOTS */
} silo_gpu_bitprog;

/* out_bitset_dev (Wp words) and/or out_count_dev (SILO_GPU_COUNT_SHARDS uint64, ACCUMULATED into, so the
 * caller can sum partitions like aggregated.cpp:58-66) may be NULL. */
int silo_gpu_filter_eval(
   const silo_gpu_store* store, const silo_gpu_bitprog* program,
   uint64_t* out_bitset_dev, uint64_t* out_count_dev, void* stream
);

/* K3 for a batch of programs over one store in ONE launch: the filter -> Aggregated queries that are in flight at the
 * same time (the reference evaluates each on its own request thread: intersection.cpp:111-126, union.cpp:39-44,
 * threshold.cpp:93-128 once per request).  A single 32-column program is launch-latency bound (40 MB at 10 M
 * sequences); a batch streams at memory speed.  out_bitsets_dev (may be NULL, entries may be NULL): per program a
 * row-sized device bitset to receive the result; out_counts (host memory, may be NULL): the cardinalities.
 * Synchronises `stream` before it returns.  Programs obey the limits of silo_gpu_filter_eval. */
int silo_gpu_filter_eval_batch(
   const silo_gpu_store* store, const silo_gpu_bitprog* programs, uint32_t n_programs, uint64_t* const* out_bitsets_dev, uint64_t* out_counts,
   void* stream
);

/* ---- K2: cardinality (aggregated.cpp:61, mutations.cpp:45) ---------------------------------------- */
/* Count slot: the cardinality of a filter without a copy, a device-side reduction or a stream synchronisation.  Every block
 * of the K3 launch stores the rows IT selected into page-locked host memory (one posted 8-byte system-scope store, tagged
 * with the launch's epoch); silo_gpu_count_slot_wait spins on those words and adds them up (after a spin budget of tens of
 * milliseconds it synchronises the stream once, so a failed launch is an error, not a hang).  One slot serves one launch at
 * a time; a host thread keeps its own.  Replaces roaring::cardinality() at the end of Operator::evaluate for Aggregated
 * (aggregated.cpp:61). */
typedef struct silo_gpu_count_slot silo_gpu_count_slot;
int silo_gpu_count_slot_create(silo_gpu_count_slot** out_slot);
void silo_gpu_count_slot_destroy(silo_gpu_count_slot* slot);
int silo_gpu_filter_eval_count(
   const silo_gpu_store* store, const silo_gpu_bitprog* program, uint64_t* out_bitset_dev /* may be NULL */, silo_gpu_count_slot* slot,
   void* stream
);
int silo_gpu_count_slot_wait(silo_gpu_count_slot* slot, uint64_t* out_count, void* stream);
int silo_gpu_popcount(const silo_gpu_store* store, const uint64_t* bitset_dev, uint64_t* out_count_dev /* shards, accumulated */, void* stream);

/* ---- K5 / K6: metadata columns -------------------------------------------------
 * A column is a plain device array with one value per row: int32 (int columns, NULL = INT32_MIN), uint32 (dates as
 * year<<16|month<<12|day, NULL = 0; dictionary ids of string / lineage columns) or double (NULL = NaN).
 * silo_gpu_bitset_from_compare evaluates one predicate of the reference's Selection operator for all rows:
 * bit i = values[i] <comparator> *value with the C++ comparison operators (CompareToValueSelection<T>::match,
 * selection.cpp:145-165) — a metadata predicate enters the fused filter program as this bitset. */
#define SILO_GPU_VALUE_I32 0
#define SILO_GPU_VALUE_U32 1
#define SILO_GPU_VALUE_F64 2
#define SILO_GPU_CMP_EQUALS 0
#define SILO_GPU_CMP_NOT_EQUALS 1
#define SILO_GPU_CMP_LESS 2
#define SILO_GPU_CMP_HIGHER_OR_EQUALS 3
#define SILO_GPU_CMP_HIGHER 4
#define SILO_GPU_CMP_LESS_OR_EQUALS 5
int silo_gpu_upload_column(const void* src_host, size_t n_rows, int value_type, void** out_dev); /* free: silo_gpu_free */
int silo_gpu_bitset_from_compare(
   const silo_gpu_store* store, uint64_t* dst_dev, const void* values_dev, int value_type, int comparator,
   const void* value /* one int32 / uint32 / double on the host */, void* stream
);
/* Aggregated with groupByFields (aggregated.cpp:100-149) for columns given as dictionary ids: for every row i of
 * the filter (NULL = all rows) counts_dev[sum_c ids[c][i] * stride_c] += 1 with mixed-radix strides, the first
 * column most significant; prod(cardinalities) <= SILO_GPU_MAX_GROUP_BINS entries, accumulated into (zero them
 * first; partitions with a shared dictionary may share one table). */
#define SILO_GPU_MAX_GROUP_COLUMNS 8
#define SILO_GPU_MAX_GROUP_BINS (1u << 24)
int silo_gpu_group_count(
   const silo_gpu_store* store, const uint64_t* filter_dev, const uint32_t* const* group_ids_dev, const uint32_t* cardinalities,
   uint32_t n_columns, uint32_t* counts_dev, void* stream
);
/* The same group-by for tuple spaces beyond SILO_GPU_MAX_GROUP_BINS (up to 2^64 - 1 potential tuples): a hash table
 * in HBM keyed by the 64-bit mixed-radix tuple id, then compacted.  max_rows bounds the number of selected rows (the
 * filter's cardinality; the table gets twice as many slots).  On success *out_keys_dev / *out_counts_dev hold
 * *out_n_groups (tuple id, count) pairs in no particular order; free both with silo_gpu_free.  Synchronises. */
int silo_gpu_group_count_hashed(
   const silo_gpu_store* store, const uint64_t* filter_dev, const uint32_t* const* group_ids_dev, const uint32_t* cardinalities,
   uint32_t n_columns, uint32_t max_rows, uint64_t** out_keys_dev, uint32_t** out_counts_dev, uint32_t* out_n_groups, void* stream
);
/* Insertion index on the device (insertion_index.cpp, insertions.cpp:186-221).  The occurrences of a column's
 * insertions in one sequence are n_pairs pairs (rows_dev[k], ids_dev[k]): row k carries the distinct insertion
 * ids_dev[k] (< n_ids).  silo_gpu_bitset_from_pairs: dst = rows of the pairs whose insertion is a member
 * (membership_by_id on the host: the distinct insertions the search pattern matched, InsertionContains);
 * silo_gpu_count_pairs: counts_dev[id] += number of pairs of that insertion whose row is in the filter (NULL = all
 * rows), the and_cardinality per insertion of the Insertions action. */
int silo_gpu_bitset_from_pairs(
   const silo_gpu_store* store, uint64_t* dst_dev, const uint32_t* rows_dev, const uint32_t* ids_dev, uint32_t n_pairs,
   const uint8_t* membership_by_id, uint32_t n_ids, void* stream
);
int silo_gpu_count_pairs(
   const silo_gpu_store* store, const uint64_t* filter_dev, const uint32_t* rows_dev, const uint32_t* ids_dev, uint32_t n_pairs,
   uint32_t* counts_dev, void* stream
);
/* FastaAligned (fasta_aligned.cpp:44-83 reconstructSequence): the stored symbol of every position of the given
 * rows as characters, out_chars_dev[r * positions + p]; row_ids_dev holds n_rows sequence ids of this store. */
int silo_gpu_reconstruct_sequences(
   const silo_gpu_store* store, uint32_t seqstore_id, const uint32_t* row_ids_dev, uint32_t n_rows, char* out_chars_dev, void* stream
);

/* ---- K4: row selection of Mutations (mutations.cpp:184-232) ------------------------------------------
 * For every position p < n_positions with total = sum_s counts[p][s] > 0, every symbol index s != reference_index[p]
 * (0xFF = the reference symbol is not a valid mutation symbol) with
 *     counts[p][s] > (min_proportion == 0 ? 0 : (uint32_t)(ceil((double)total * min_proportion) - 1))
 * is appended to the list in out_dev: word 0 = number of selected cells (may exceed `capacity`: then only the
 * first `capacity` appended rows are stored and the caller falls back to the whole table), words 1-3 unused,
 * rows from word 4.  Rows are unordered.  out_dev holds 16 + 16 * capacity bytes. */
typedef struct silo_gpu_mutation_row {
   uint32_t position;     /* 0-based, relative to counts_dev */
   uint32_t symbol_index; /* index into the sequence store's valid mutation symbols */
   uint32_t count;
   uint32_t total;
} silo_gpu_mutation_row;
int silo_gpu_mutations_select(
   const uint32_t* counts_dev, const uint8_t* reference_index_dev, uint32_t n_positions, uint32_t n_symbols, double min_proportion,
   uint32_t capacity, uint32_t* out_dev, void* stream
);
/* K4 with the list delivered straight into page-locked host memory: a row slot owns a host buffer the kernel writes the
 * selected rows into and a header word its last block publishes (no device -> host copy, no event: the host spins on the
 * word as for a count slot).  One slot serves one launch at a time.  *out_selected may exceed the slot's capacity: only the
 * first `capacity` rows are there and the caller falls back to the whole table, as with silo_gpu_mutations_select.
 * The rows stay valid until the slot's next launch. */
typedef struct silo_gpu_row_slot silo_gpu_row_slot;
int silo_gpu_row_slot_create(uint32_t row_capacity, silo_gpu_row_slot** out_slot);
void silo_gpu_row_slot_destroy(silo_gpu_row_slot* slot);
int silo_gpu_mutations_select_to_slot(
   const uint32_t* counts_dev, const uint8_t* reference_index_dev, uint32_t n_positions, uint32_t n_symbols, double min_proportion,
   silo_gpu_row_slot* slot, void* stream
);
int silo_gpu_row_slot_wait(silo_gpu_row_slot* slot, const silo_gpu_mutation_row** out_rows, uint32_t* out_selected, void* stream);
/* Plain byte upload into a fresh device allocation (free with silo_gpu_free). */
int silo_gpu_upload_bytes(const void* src_host, size_t bytes, void** out_dev);

/* ---- K1: Mutations scan (mutations.cpp:64-164) ---------------------------------------------------
 * counts_out_dev[(p - pos_begin) * n_scan_symbols + s] += popcount(filter & {rows whose symbol at p is scan_symbols[s]})
 * for p in [pos_begin, pos_end), decoded from the bit-sliced scan planes (3 planes per nucleotide position, 5 per
 * amino-acid position; the scan symbols must be the 5 / 22 valid mutation symbols).  The buffer is ACCUMULATED into (the reference sums partitions into
 * one table, mutations.cpp:71,108); zero it with silo_gpu_memset_async before the first partition.
 * filter_dev == NULL means the full filter: like the reference, which then reads stored cardinalities
 * instead of intersecting (mutations.cpp:98-136), the totals of the unfiltered store are computed by one
 * scan on first use, kept on the device and added from then on (invalidated by append / generate).
 * A filter whose set bits fall into few 64-byte sectors (<= row_words / 16 by default, SILO_GPU_TUNE_SCAN_SPARSE_DIVISOR; and
 * fewer than the column tiles the dense scan cannot skip are worth: a clustered filter stays with the dense scan)
 * is served by a gather over just those sectors of the planes (K1s) — decided on the device, same counts, so that the
 * scan gets cheaper with the filter as roaring's and_cardinality does (mutations.cpp:139-164). */
int silo_gpu_mutations_scan(
   const silo_gpu_store* store, uint32_t seqstore_id, const uint64_t* filter_dev,
   uint32_t pos_begin, uint32_t pos_end, uint32_t* counts_out_dev, void* stream
);
/* K1i, the compact scan index.  silo_gpu_store_finalize derives, per sequence store, two code planes per position (code
 * 1..3 = the three most frequent valid symbols AT THAT POSITION, 0 = anything else) plus the few rows whose valid symbol
 * is none of the three as explicit keys; the Mutations scan streams those 2 planes instead of the 3 / 5 full code planes
 * and adds the exceptions in one small pass — same counts.  Built only when the exceptions stay below 1/512 (nucleotides) or 1/170 (amino acids) of the cells
 * and the device memory is there (otherwise, and for every other consumer, the full planes serve).
 * silo_gpu_store_scan_planes: plane rows the scan reads per position (2 with the index, else 3 / 5);
 * silo_gpu_store_scan_escapes: number of exception keys (0 without the index). */
uint32_t silo_gpu_store_scan_planes(const silo_gpu_store* store, uint32_t seqstore_id);
uint64_t silo_gpu_store_scan_escapes(const silo_gpu_store* store, uint32_t seqstore_id);
/* Plane rows (of Wp words each) the Mutations scan reads for positions [pos_begin, pos_end) of a sequence store: the
 * physical bytes of a scan are this x 8 Wp, plus the filter and 8 bytes per escape key. */
uint64_t silo_gpu_store_scan_rows(const silo_gpu_store* store, uint32_t seqstore_id, uint32_t pos_begin, uint32_t pos_end);
/* Derived symbols.  At almost every position ONE valid symbol has nearly every row; the reference leaves that symbol's bitmap
 * out and rebuilds its count as |filter| - #missing - the other symbols' counts (position.cpp:102-127, mutations.cpp:74-95).
 * A finalized store whose missing symbol is kept as runs does the same: such a position stores no row and no key for its most
 * numerous symbol, and a scan derives that count from the filter's cardinality, the rows of the filter inside a run of the
 * missing symbol or with an ambiguity code, and the other symbols' counts — the same numbers.  silo_gpu_store_scan_planes is
 * then 0 (most positions have no row at all); a scan additionally reads the runs (12 bytes each) and the sparse keys (8 bytes
 * each): silo_gpu_store_scan_runs / _scan_sparse_keys (0 for a store without derived symbols). */
uint64_t silo_gpu_store_scan_runs(const silo_gpu_store* store, uint32_t seqstore_id);
uint64_t silo_gpu_store_scan_sparse_keys(const silo_gpu_store* store, uint32_t seqstore_id);
/* Finalizes ONE sequence store (silo_gpu_store_finalize does all that are left): its build-time planes are re-encoded into
 * the adaptive code planes and released.  A loader that fills the stores of a partition one after the other calls this
 * after each, so that their build-time planes are never resident together (10 M sequences: 112 GB for the nucleotide
 * genome alone).  No sequences can be appended to a finalized store. */
int silo_gpu_store_finalize_seqstore(silo_gpu_store* store, uint32_t seqstore_id);

/* K1 over several position ranges at once — the 12 genes of an AminoAcidMutations query, the segments of a segmented
 * genome, the sequence stores of several batched queries: every filter is applied to every range;
 * counts_out_dev[r * n_filters + q] is the table of filter q on range r, indexed from the range's first position and
 * ACCUMULATED into.  Ranges over stores of the same alphabet share launches (blocks are dealt to the ranges), and the
 * sparse-filter routing (K1s) looks at each filter once for all ranges.  Replaces the reference's loop over sequence
 * names in Mutations::execute (mutations.cpp:258-271). */
typedef struct silo_gpu_scan_range {
   uint32_t seqstore_id;
   uint32_t pos_begin;
   uint32_t pos_end;
} silo_gpu_scan_range;
int silo_gpu_mutations_scan_ranges(
   const silo_gpu_store* store, const silo_gpu_scan_range* ranges, uint32_t n_ranges, const uint64_t* const* filters_dev, uint32_t n_filters,
   uint32_t* const* counts_out_dev, void* stream
);

/* The same scan for a batch of filters over one sequence store: every plane row is read once for up to
 * SILO_GPU_MAX_SCAN_BATCH filters per pass (larger batches take several passes), counts_out_dev[q] is
 * accumulated with filters_dev[q].  This is how concurrent Mutations queries share the HBM stream. */
enum { SILO_GPU_MAX_SCAN_BATCH = 8 };
int silo_gpu_mutations_scan_batch(
   const silo_gpu_store* store, uint32_t seqstore_id, const uint64_t* const* filters_dev, uint32_t n_filters,
   uint32_t pos_begin, uint32_t pos_end, uint32_t* const* counts_out_dev, void* stream
);
int silo_gpu_memset_async(void* dev_ptr, int value, size_t bytes, void* stream);

/* Tuning knobs of K1 (0 = default); returns the previous value.  For benchmarks only. */
enum { SILO_GPU_TUNE_SCAN_ROWS_PER_BLOCK = 0, SILO_GPU_TUNE_SCAN_VARIANT = 1, SILO_GPU_TUNE_EVAL_LEAF_BATCH = 2 /* 8 (default) or 16 leaf loads in flight per lane in K3 */,
       SILO_GPU_TUNE_COMPACT_INDEX = 4 /* finalize: < 0 keeps the build-time identity planes, 0 (default) re-encodes every position into its cheapest
                                          layout (one-hot rows with the most numerous symbol derived, 2 / 3 code planes, identity planes), 2 the same
                                          without one-hot rows, 3 with a one-hot row for the most numerous symbol too (nothing derived) */,
       SILO_GPU_TUNE_KEY_COST = 6 /* finalize: > 0 = the cost of an escape key, in plane bytes, in the choice of layouts (experiments) */,
       SILO_GPU_TUNE_MISSING_RUNS = 8 /* finalize: < 0 keeps the plane of the missing symbol (N / X) instead of turning it into runs */,
       SILO_GPU_TUNE_LAUNCH_COST = 9 /* finalize: what a further kind of plane-scan launch costs in the choice of layouts, in KiB of plane bytes; 0 = default (192 MiB), < 0 = nothing (small test stores that are to mix layouts) */,
       SILO_GPU_TUNE_SCAN_TIMING = 7 /* 1: bracket every plane-scan launch with HIP events (silo_gpu_scan_timings) */,
       SILO_GPU_TUNE_SIDE_STREAM = 5 /* the escape-key pass of a scan: 0 (default) on a side stream of the lowest priority, 1 of default priority, 2 on the caller's stream, 3 = as 0 over the position-major keys (k_scan_escapes) */,
       SILO_GPU_TUNE_SCAN_SPARSE_DIVISOR = 3 /* a filter with a set bit in <= row_words / divisor of its 64-byte sectors takes the gather scan (K1s); 0 = default 16, < 0 = off */ };
int silo_gpu_tune(int knob, int value);

/* HIP events on the caller's stream, so a host without the HIP headers can time the kernels
 * (bench.py measures the roofline numbers with these, on the stream the kernels are launched on). */
int silo_gpu_event_create(void** out_event);
int silo_gpu_event_record(void* event, void* stream);
int silo_gpu_event_synchronize(void* event); /* blocks the calling thread until the event has happened */
int silo_gpu_event_elapsed_ms(void* start_event, void* stop_event, float* out_ms); /* synchronises on stop */
void silo_gpu_event_destroy(void* event);

/* ---- exchange step of the sharded scan: RCCL over xGMI -------------------------------------
 * One process per GPU.  The reference sums its partitions in a host loop inside one process (query_engine.cpp:40-49,
 * mutations.cpp:71,108); across GPUs that sum is ONE all-reduce of the count table per query (position-range or
 * sequence-id shards alike), and a filter leaf at a position another rank holds is ONE broadcast of a row bitset.
 * silo_gpu_comm_unique_id: called by one rank; the SILO_GPU_COMM_ID_BYTES opaque bytes reach the other ranks out
 * of band (the launcher's key-value store, MPI, a file).  silo_gpu_comm_create: collective over all `world` ranks.
 * The collectives are enqueued on `stream` (any stream of the communicator's device; NULL = the null stream) and
 * ordered with the kernels on it — no host synchronisation.  Every rank must enqueue the same collectives in the same
 * order; calls on one communicator are serialised internally.  librccl is bound on first use: without it these
 * fail with SILO_GPU_ERR_UNSUPPORTED and nothing else in the library is affected. */
typedef struct silo_gpu_comm silo_gpu_comm;
enum { SILO_GPU_COMM_ID_BYTES = 128 };
int silo_gpu_comm_unique_id(uint8_t* out_id /* [SILO_GPU_COMM_ID_BYTES] */);
int silo_gpu_comm_create(const uint8_t* id, uint32_t rank, uint32_t world, int device, silo_gpu_comm** out);
void silo_gpu_comm_destroy(silo_gpu_comm* comm);
uint32_t silo_gpu_comm_rank(const silo_gpu_comm* comm);
uint32_t silo_gpu_comm_world(const silo_gpu_comm* comm);
/* counts_dev[i] = sum over ranks of counts_dev[i], in place (ncclAllReduce, ncclUint32, ncclSum). */
int silo_gpu_allreduce_counts(silo_gpu_comm* comm, uint32_t* counts_dev, size_t n, void* stream);
/* bytes_dev of rank `root` to every rank, in place (ncclBroadcast). */
int silo_gpu_broadcast_bytes(silo_gpu_comm* comm, void* bytes_dev, size_t n_bytes, uint32_t root, void* stream);

/* Name of the last kernel variant silo_gpu_mutations_scan launched (for roofline attribution). */
const char* silo_gpu_last_scan_kernel(void);

/* Per-launch timing of a scan (measurement only).  With SILO_GPU_TUNE_SCAN_TIMING set to 1, every k_scan_sliced,
 * k_scan_escapes_sliced, k_scan_missing_runs and k_count_sparse_keys launch of the calling thread's scans is bracketed by HIP
 * events on the stream it is launched on; this call waits for the
 * events of that thread's LAST scan and returns one entry per launch (at most `capacity`; *n_out = launches).  plane_rows =
 * plane rows the launch streams (each once, row_words * 8 bytes), filters = filter rows it holds in registers. */
typedef struct silo_gpu_scan_timing {
   char kernel[64];       /* e.g. "k_scan_sliced<2, 2, 8, 1, 2>", as rocprofv3 names it */
   uint64_t plane_rows;   /* plane rows a k_scan_sliced launch streams (0 for the other kernels) */
   uint64_t bytes;        /* what the launch has to read, each byte once: plane rows + filter rows; 8 bytes per escape key (+ a filter
                             slice per block); 12 bytes per run of the missing symbol; 8 per sparse key */
   uint32_t filters;
   uint32_t blocks;
   float ms;
} silo_gpu_scan_timing;
int silo_gpu_scan_timings(silo_gpu_scan_timing* out, uint32_t capacity, uint32_t* n_out);

/* The achievable HBM read rate of this device (measurement only; SURVEY.md section 8(d): "also measure an achievable-stream-read
 * ceiling with a plain uint64 sum kernel"): allocates `bytes` of device memory, sums it `reps` times with 16-byte non-temporal
 * loads (8 in flight per lane) and returns the average duration of one pass in *out_ms_per_pass (HIP events, null stream). */
int silo_gpu_stream_read_probe(uint64_t bytes, uint32_t reps, float* out_ms_per_pass);

/* Thread-local description of the last error. */
const char* silo_gpu_last_error(void);

#ifdef __cplusplus
}
#endif
#endif /* SILO_GPU_H */
