/*
 *  kernels.h - the thin C layer between the C host (csrc/host) and the hand-written gfx950 kernels (csrc/hip).
 *
 *  Everything here is `extern "C"`, takes plain pointers and sizes, enqueues work on the given HIP stream and
 *  returns the `hipError_t` of the launch as an int (0 = success).  No call blocks; no call allocates.
 *
 *  Data model shared by all kernels:
 *    - a string is a `szs_string_ref_t` {absolute device address, byte length, index in the caller's collection};
 *    - candidates arrive LENGTH-SORTED so that the 64 lanes of a wavefront, one candidate each, walk texts of
 *      near-equal length (lock-step waste is bounded by the length spread inside one 64-candidate block);
 *    - queries arrive grouped by kernel variant; a workgroup scores ONE query against 256 candidates;
 *    - results go straight to `results[query.index * stride + candidate.index]` as 64-bit values (plus the
 *      mirror cell in symmetric mode) - there is no per-cell task array, no sort of tasks, no scatter pass
 *      (the reference's cuda.cuh:1652-1711,2082,2146 machinery has no counterpart here).
 */
#ifndef SZS_ROCM_KERNELS_H_
#define SZS_ROCM_KERNELS_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct szs_string_ref_t {
    uint64_t address; /* absolute, device-accessible */
    uint32_t length;  /* bytes */
    uint32_t index;   /* row (queries) or column (candidates) of the results matrix */
} szs_string_ref_t;

/* The `symmetric` argument of every scoring launcher is a set of layout flags: */
#define SZS_LAYOUT_SYMMETRIC 1  /* score candidate.index <= query.index only and mirror each result */
#define SZS_LAYOUT_TRANSPOSED 2 /* the host swapped the sides: results[candidate.index * stride + query.index] */

#define SZS_CANDIDATES_PER_WORKGROUP 256u
#define SZS_MYERS_MAX_WORDS 64u /* 32-bit words: queries up to 2048 bytes take the bit-parallel kernel */
#define SZS_MYERS_SHORT_WORDS 8u /* queries up to 256 bytes share ONE launch that picks the width per workgroup */

/**
 *  Unit-cost Levenshtein, bit-parallel Myers/Hyyro on 32-bit words with a full-width carry chain.
 *  `words` is a launch variant from szs_hip_levenshtein_myers_round_words(): SZS_MYERS_SHORT_WORDS scores any mix of
 *  queries of up to 256 bytes (each at its own exact width); a larger value requires ceil(length / 32) <= words for
 *  EVERY query in `queries[0..queries_count)`.  Queries should arrive longest first (heaviest workgroups first).
 *  Launches ceil(candidates_count / 256) * queries_count workgroups of 256 threads.
 *  symmetric != 0: only cells with candidate.index <= query.index are scored, and mirrored.
 */
unsigned szs_hip_levenshtein_myers_round_words(unsigned words);

/**
 *  Refs that were planned for an EARLIER call may be handed to the byte kernels again when the caller passes the same tapes
 *  (same data and offsets pointers, same counts): with a guard, every workgroup checks its query ref and every lane its
 *  candidate ref against the offsets AS THEY ARE NOW - `offsets[index + 1] - offsets[index] == length` and
 *  `base + offsets[index] == address` - BEFORE touching the string.  A ref that no longer describes its string is not
 *  dereferenced, nothing is written for it, and `*stale` (pinned host memory) receives `sequence`; the host then re-plans.
 *  `side[0]` describes the tape of the kernel's queries, `side[1]` of its candidates.
 */
typedef struct szs_ref_guard_t {
    uint32_t enabled, sequence;
    uint32_t *stale;
    struct {
        void const *offsets;
        uint64_t base;
        uint32_t wide, count;
    } side[2];
} szs_ref_guard_t;

int szs_hip_levenshtein_myers(unsigned words, szs_string_ref_t const *queries, uint32_t queries_count,
                              szs_string_ref_t const *candidates, uint32_t candidates_count, uint64_t *results,
                              uint64_t results_row_stride, int symmetric, szs_ref_guard_t const *guard /* may be NULL */,
                              void *stream);

/**
 *  The same for the long widths (`words` = 24 (2 lanes only), 32, 48 or 64) with every pair spread over `lanes` = 2 or 4 adjacent lanes - a
 *  strip pipeline inside the wavefront: 1 / lanes of the one-lane-per-pair kernels' floor (the longest pair), the same work.
 *  Every query of the launch fits `words`; a workgroup scores one query against 256 / lanes candidates.
 */
int szs_hip_levenshtein_myers_split(unsigned words, unsigned lanes, szs_string_ref_t const *queries, uint32_t queries_count,
                                    szs_string_ref_t const *candidates, uint32_t candidates_count, uint64_t *results,
                                    uint64_t results_row_stride, int symmetric, szs_ref_guard_t const *guard /* may be NULL */,
                                    void *stream);

/**
 *  Codepoint-level twin of the short-query bit-parallel kernel: strings are UTF-32 arrays (`address` points at `u32`
 *  runes, `length` counts runes) produced by szs_hip_utf8_transcode; every query has at most 256 runes.  This launcher and
 *  the two below read a candidate's text 16 bytes at a time: every array must START ON A 16-BYTE BOUNDARY and own its
 *  storage up to the next one (dispatch.c pads the rune offsets accordingly).
 */
int szs_hip_levenshtein_myers_runes(szs_string_ref_t const *queries, uint32_t queries_count,
                                    szs_string_ref_t const *candidates, uint32_t candidates_count, uint64_t *results,
                                    uint64_t results_row_stride, int symmetric, uint32_t alphabet /* 0, or the size of the batch's renumbered alphabet (szs_hip_alphabet_rename) */,
    void *stream);

/**
 *  Byte queries of more than 2048 bytes, unit costs: the same recurrence in horizontal strips of 2048 rows; the deltas
 *  under a strip's last row (one bit pair per text column) are parked in `workspace` - szs_hip_levenshtein_myers_banded_bytes()
 *  bytes, sized by the RESIDENT workgroups - for the strip below.  Queries should arrive longest first.
 */
int szs_hip_levenshtein_myers_banded(szs_string_ref_t const *queries, uint32_t queries_count, szs_string_ref_t const *candidates,
                                     uint32_t candidates_count, uint32_t longest_candidate, uint64_t *results,
                                     uint64_t results_row_stride, int symmetric, void *workspace, void *stream);
size_t szs_hip_levenshtein_myers_banded_bytes(uint32_t queries_count, uint32_t candidates_count, uint32_t longest_candidate);

/**
 *  The long rune widths (16, 24 (2 lanes only), 32, 48, 64 words) with every pair spread over `lanes` = 2 or 4 adjacent lanes and
 *  `pairs_per_workgroup` pairs - that many x lanes threads - per workgroup: the rune table of a 48- or 64-word query fills a CU's LDS, and this way
 *  it serves `lanes` wavefronts per SIMD instead of one.  hipErrorNotSupported like szs_hip_levenshtein_myers_runes_long.
 */
int szs_hip_levenshtein_myers_runes_split(unsigned words, unsigned lanes, szs_string_ref_t const *queries, uint32_t queries_count,
                                          szs_string_ref_t const *candidates, uint32_t candidates_count, uint64_t *results,
                                          uint64_t results_row_stride, int symmetric, uint32_t alphabet /* 0, or the size of the batch's renumbered alphabet (szs_hip_alphabet_rename) */,
                                          uint32_t pairs_per_workgroup /* 256, or 64 for a launch of a few workgroups */, void *stream);

/**
 *  Codepoint queries of 257 to 2048 runes: `words` is a long launch variant of szs_hip_levenshtein_myers_round_words()
 *  (10 ... 64) and every query of the launch fits it.  Peq rows are keyed by dense rune ids in dynamic LDS (up to 160 KB
 *  per workgroup); runes beyond the table's capacity are matched against the pattern directly, so the result is exact
 *  whatever the alphabet.  Returns hipErrorNotSupported when the device refuses that much LDS - the caller then scores
 *  the group with the rune-keyed DP kernel (szs_hip_weighted_scores, szs_objective_distance_runes_k).
 */
int szs_hip_levenshtein_myers_runes_long(unsigned words, szs_string_ref_t const *queries, uint32_t queries_count,
                                         szs_string_ref_t const *candidates, uint32_t candidates_count, uint64_t *results,
                                         uint64_t results_row_stride, int symmetric, uint32_t alphabet /* 0, or the size of the batch's renumbered alphabet (szs_hip_alphabet_rename) */,
    void *stream);

/**
 *  Codepoint queries of more than 2048 runes, unit costs: the strips of szs_hip_levenshtein_myers_banded with a dense-id
 *  rune table rebuilt per strip (dynamic LDS, like szs_hip_levenshtein_myers_runes_long).  `workspace` holds
 *  szs_hip_levenshtein_myers_banded_runes_bytes() bytes (0: the device cannot host the table).  Returns hipErrorNotSupported
 *  when the device refuses the LDS - the caller then scores the group with the rune-keyed DP kernel.
 */
int szs_hip_levenshtein_myers_banded_runes(szs_string_ref_t const *queries, uint32_t queries_count, szs_string_ref_t const *candidates,
                                           uint32_t candidates_count, uint32_t longest_candidate, uint64_t *results,
                                           uint64_t results_row_stride, int symmetric, void *workspace, void *stream);
size_t szs_hip_levenshtein_myers_banded_runes_bytes(uint32_t queries_count, uint32_t candidates_count, uint32_t longest_candidate);

/**
 *  ONE persistent launch for a unit-cost byte-level call whose queries span several bit-vector widths (hip/myers_queue.hip):
 *  work items (`queries_per_item` consecutive queries, `candidates_per_item` consecutive candidates) are drawn from a ticket
 *  counter in the order the host planned (host/plan.c: szs_plan_queue) - TILES of (a slice of the queries, longest first) x (a
 *  column of the candidates, ascending), sorted by how long one of their items holds a workgroup.  Inside a tile the blocks of
 *  a column are taken from its end (heaviest first), every group of queries of the slice against one block before the next.
 *  A tile's shape: `lanes` = 1 scores a pair on one lane at the query's own width (queries of up to 16 words = 512 bytes);
 *  `lanes` = 2 ... 16 spreads it over that many adjacent lanes of `words_per_lane` = 4, 8, 12 or 16 words.  A query that does
 *  not fit its tile's shape is scored at a shape that takes it (the kernel never trusts the plan with correctness).
 */
#define SZS_QUEUE_MOST_TILES 96u
typedef struct szs_queue_tile_t {
    uint32_t first_item;                     /* items of all tiles before this one */
    uint32_t query_first, query_count;       /* slice of the query refs (longest first) */
    uint32_t candidate_first, candidate_end; /* column of the candidate refs (ascending) */
    uint32_t candidates_per_item;            /* S: the masks of a query are built once per S candidates */
    uint8_t words_per_lane, lanes;
    uint8_t queries_per_item;                /* G <= 16: the masks of G queries side by side in the workgroup's table (each gets 1 / G
                                                of it); the slice's last item may hold fewer */
    uint8_t flags;                           /* SZS_QUEUE_TILE_SPARSE */
} szs_queue_tile_t;
#define SZS_QUEUE_TILE_SPARSE 1u /* codepoints: the tile's tables are pointers + a pool of non-zero chunks (hip/myers_queue.hip) */
typedef struct szs_queue_plan_t {
    uint32_t tiles_count, items_total;
    uint32_t chain_most; /* the longest chain of dependent steps a wave block of this queue holds: words one lane holds x the longest
                            candidate of its column (0: wave blocks all run at one priority) */
    szs_queue_tile_t tiles[SZS_QUEUE_MOST_TILES];
} szs_queue_plan_t;
/**
 *  `tickets`: one dword of device memory, zeroed when allocated and never again; `ticket_base`: its value when this launch
 *  begins; `*tickets_taken` receives what the launch adds to it (items + workgroups: every workgroup stops at the first
 *  ticket past the end), so the caller tracks the counter without reading it back.
 */
int szs_hip_levenshtein_myers_queue(szs_queue_plan_t const *plan, szs_string_ref_t const *queries, szs_string_ref_t const *candidates,
                                    uint64_t *results, uint64_t results_row_stride, int layout, uint32_t *tickets, uint32_t ticket_base,
                                    uint32_t *tickets_taken, uint64_t *trace /* NULL, or 7 x szs_hip_levenshtein_myers_queue_grid(items)
                                    qwords of device memory: per workgroup its first and last 100 MHz tick, the items it took, its last item and when that
                                    began, its longest item and how long that took */,
                                    uint32_t alphabet /* 0: byte strings; A: UTF-32 arrays of ids 1 ... A (szs_hip_alphabet_rename) */,
                                    uint32_t *unfit_flag /* pinned host memory, or NULL: receives `unfit_sequence` when a query fits no
                                    table of the kernel (the plan's job to prevent; the host then scores the call another way) */,
                                    uint32_t unfit_sequence, void *stream);
unsigned szs_hip_levenshtein_myers_queue_grid(uint64_t items, int runes);
/** Bytes of LDS a workgroup's tables share: 64 KB for bytes, 72 KB for codepoints (what szs_plan_queue fits its tiles into). */
size_t szs_hip_levenshtein_myers_queue_table_bytes(int runes);

/**
 *  Unit-cost byte-level distances over TINY strings (hip/myers_tiny.hip), straight from the tapes - no refs, no planner, ONE launch:
 *  a workgroup scores 256 consecutive candidates against a span of the queries, THIRTY-TWO queries at a time per lane (16-bit
 *  bit-vectors, two to a register), and writes whole 2 KB runs of the result rows.  Plain layout only
 *  (results[query * stride + candidate]).  The few strings of 17 ... 255 bytes are scored by the same workgroups in the shadow of
 *  that work: a block's long candidates under the groups' own masks, a span's long queries as W-word patterns over the block.
 *  Malformed offsets, a string beyond 255 bytes, or a block of candidates / span of queries of which more than a quarter is long
 *  leave `*unfit = unfit_sequence` in pinned host memory: the caller then scores the call the ordinary way.  `symbols_out` (pinned, or NULL): [0] the bytes of the queries' tape, [1] of the candidates' - their
 *  product is the call's cells.
 */
#define SZS_TINY_LONGEST 255u /* bytes of the longest string that launch scores (a distance fits a byte of its staging rows) */
typedef struct szs_tape_t {
    void const *offsets;
    uint64_t base; /* address of the tape's bytes */
    uint32_t count, wide; /* wide: 0 - 32-bit offsets, 1 - 64-bit offsets, 2 - the words szs_hip_utf8_narrow writes (below) */
} szs_tape_t;
int szs_hip_levenshtein_tiny(szs_tape_t const *queries, szs_tape_t const *candidates, uint64_t *results, uint64_t results_row_stride,
                                   uint32_t *unfit, uint32_t unfit_sequence, unsigned long long *symbols_out,
                                   uint64_t *rune_totals /* NULL, or (tapes of kind 2) the two totals szs_hip_utf8_narrow counted: read, and zeroed for the next call */,
                                   uint64_t *squares_out /* NULL, or (a symmetric call: both tapes the same) a qword per block of 256 strings in PINNED memory:
                                                            the sums of their squared lengths */,
                                   uint64_t *trace /* NULL, or 10 qwords per workgroup of device memory (`trace` knob) */,
                                   uint64_t trace_workgroups /* workgroups `trace` has room for: a larger grid is not traced at all */,
                                   int dense /* testing: score blocks and spans full of long strings too (slowly) instead of refusing them */, void *stream);

/**
 *  The CODEPOINT twin of that launch (round 6; reference: unit_utf8_per_cuda_thread_, cuda.cuh:3294 - short runes scored without the
 *  general codepoint machinery): one pass (hip/utf8.hip: utf8_narrow_kernel) decodes both UTF-8 tapes - `sz_rune_decode_unchecked`'s
 *  contract, as the transcoder - and writes every string as BYTES, one per rune: an ASCII rune is its own id, any other rune gets
 *  128 + its slot in a table of SZS_NARROW_SLOTS claimed runes.  Only equality of symbols matters to the distance, so the byte
 *  kernel then scores the narrow strings as they are.  `entries[i]` (queries first, then candidates): string i's place in `narrow`
 *  below bit 56, its count of runes above - a szs_tape_t of kind 2.  `workspace`: SZS_NARROW_SLOTS dwords (the table: zeroed by the
 *  caller ONCE - it lives on from call to call, a stream of batches claims its runes in the first - and again after a batch that
 *  overflowed it) + two qwords (the totals of runes per side: zero before the pass; the scoring launch reads them and zeroes them).  Leaves `*unfit = unfit_sequence` (and the strings unscored) when the tapes'
 *  offsets descend, a string has more than 4 x 255 bytes or more than 255 runes, the batch more than SZS_NARROW_SLOTS distinct
 *  runes beyond ASCII, or the tapes more bytes than `capacity`.
 */
#define SZS_NARROW_SLOTS 128u
#define SZS_NARROW_WORKSPACE 1024u /* bytes a caller sets aside for `workspace` */
int szs_hip_utf8_narrow(szs_tape_t const *queries, szs_tape_t const *candidates, void *narrow, uint64_t capacity, uint64_t *entries,
                        void *workspace, uint32_t *unfit, uint32_t unfit_sequence, void *stream);

/**
 *  Fills the cells above the diagonal of a `side` x `side` matrix of 8-byte values from the ones below it (hip/mirror.hip): what a
 *  symmetric call sharded over several GPUs by bands of rows (host/node.c) needs once every band has landed.
 */
int szs_hip_mirror_lower(uint64_t *matrix, uint32_t side, uint64_t row_stride, void *stream);

/* ---- tuning knobs (host/tuning.c): read from the environment ONCE at load, changed only by szs_rocm_tuning_set -------- */

enum {
    szs_knob_tier_k = 0,    /* -1 automatic | 0 lanes | 1 systolic | 2 chain */
    szs_knob_swap_k,        /* -1 automatic | 0 | 1: the planner's orientation */
    szs_knob_packed_k,      /* -1 automatic | 0: pin the 32-bit weighted kernel */
    szs_knob_rune_ids_k,    /* -1 automatic | n: shrink the rune table of the long codepoint kernels to n ids */
    szs_knob_chain_waves_k, /* -1 automatic | 4 / 8 / 16: wavefronts per workgroup of the bit-parallel chain */
    szs_knob_trace_k,       /* 0 | 1: per-phase host times of every call on stderr */
    szs_knob_cells_k,       /* -1 automatic | 64: force the 64-bit cell tier */
    szs_knob_planner_k,     /* -1 automatic | 0 host | 1 device */
    szs_knob_speculate_k,   /* -1 automatic | 0: never enqueue scoring launches before the plan is known */
    szs_knob_cpu_requests_k, /* -1 / 0 strict: engines need sz_cap_cuda_k, CPU scopes are a mismatch | 1 ("gpu"): capability
                                masks without the GPU bit and CPU scopes are served by the GPU engines on device 0 */
    szs_knob_streams_k,     /* -1 automatic | 0: every launch of a call on the scope's one stream */
    szs_knob_reuse_k,       /* -1 automatic | 0: never re-use the refs planned for the previous call of the same tapes */
    szs_knob_split_k,       /* -1 automatic | 0 / 2 / 4: lanes per pair of the long byte kernels (24 ... 64 words) */
    szs_knob_alphabet_k,    /* -1 automatic | 0 never | 1 always: renumber a codepoint batch's runes (hip/utf8.hip) */
    szs_knob_merge_k,       /* -1 automatic | n: candidate blocks per workgroup of the short bit-parallel kernels (1: never merge) */
    szs_knob_team_k,        /* -1 automatic | 0: never the team tier of the 16-bit weighted scorers | lanes * 10000 + registers * 100 + waves: that shape */
    szs_knob_queues_k,      /* hardware queues the process has (GPU_MAX_HW_QUEUES when the library was loaded, else 4): the launches of a
                               call fan out over at most that many streams */
    szs_knob_roctx_k,       /* 0 | 1: the host phases of every call as roctx ranges (rocprofv3 --marker-trace) */
    szs_knob_queue_k,       /* -1 automatic (unit-cost byte calls of two or more width groups) | 0 never | 1 every unit-cost byte call:
                               the ONE persistent launch of hip/myers_queue.hip instead of a launch per width */
    szs_knob_queue_words_k, /* -1 automatic | 4 / 8 / 12 / 16: the most words of a pattern one lane may hold in that launch */
    szs_knob_queue_rounds_k,/* -1 automatic | n: candidates per work item, in rounds of the workgroup's eight wavefronts */
    szs_knob_queue_priority_k, /* -1 automatic (on) | 0: every wave block of that launch at one hardware priority | 1: longest chain first */
    szs_knob_fused_k,       /* -1 automatic | 0: never fold the planner into the short unit-cost launch (szs_fused_plan_t) | 2 (testing): its
                               sorters never publish - every waiting workgroup runs out of polls, the call is planned the ordinary way */
    szs_knob_tiny_k,        /* -1 automatic (tiny tokens on both sides) | 0 never | 1 every unit-cost byte call of strings up to 255 bytes, few
                               of them beyond 16: the tiny-token launch of hip/myers_tiny.hip | 2 (testing): dense batches are scored there too */
    szs_knob_count_k
};
int szs_tuning_get(int knob);

/* ---- the planner on the device (hip/planner.hip) ------------------------------------------------------------------------- */

#define SZS_PLAN_VARIANTS 10u /* slot 0: no bit-parallel width (weighted / strip kernels); 1..9: SZS_MYERS_SHORT_WORDS, 10 ... 64 */
#define SZS_PLAN_DEVICE_LONGEST 6143u /* longest string the device planner sorts (two LDS histograms); beyond: host planner */
#define SZS_PLAN_STATUS_DESCENDING 1u /* tape offsets do not ascend */
#define SZS_PLAN_STATUS_OVERFLOW 2u   /* a string of 4 GiB or more */
#define SZS_PLAN_STATUS_UNSORTED 4u   /* a side was not sorted (status above, or strings beyond SZS_PLAN_DEVICE_LONGEST) */

/** What the tier / orientation model (host/plan.c) needs to know about one side of a cross-product. */
typedef struct szs_side_stats_t {
    uint32_t count, longest;
    uint64_t symbols;        /* sum of the lengths */
    uint64_t bands_systolic; /* sum of ceil(length / SZS_SYSTOLIC_BAND_ROWS), at least 1 per string */
    uint64_t bands_chain;    /* the same for SZS_MYERS_CHAIN_BAND_ROWS */
} szs_side_stats_t;

/** One side of the planner's input and output: a tape (32- or 64-bit offsets, device-accessible) and two ref arrays. */
typedef struct szs_plan_side_t {
    void const *offsets;
    uint64_t base; /* address of the tape's bytes - or, with `lengths`, of the UTF-32 scratch tape */
    uint32_t count, wide;
    szs_string_ref_t *ascending, *descending;
    /* codepoint engines: string i is `lengths[i]` runes at `base + 4 * starts[i]` (szs_hip_utf8_transcode_tape wrote both);
     * the offsets are still checked (descending tapes are reported, never scored).  NULL: a byte tape. */
    uint32_t const *lengths;
    uint64_t const *starts;
} szs_plan_side_t;

/** The launch shape the host has ALREADY enqueued scoring kernels for (speculation on the previous call's shape). */
typedef struct szs_plan_expectation_t {
    uint32_t enabled;
    uint32_t query_side;  /* which side takes the kernels' query role: 0 = the caller's queries, 1 = its candidates */
    uint32_t longest[2];  /* upper bounds the workspaces and cell widths were chosen for: queries, candidates */
    uint32_t variant_counts[SZS_PLAN_VARIANTS]; /* strings per launch variant on the query side: must match exactly */
    uint32_t sequence;    /* echoed into the summary, so that a stale summary is never mistaken for this call's */
    /* codepoint calls (round 3): what the transcoder and the renumbering pass, enqueued AHEAD of the planner on the same
     * stream, left in device memory - the speculated launches were shaped for a UTF-32 buffer of `runes_capacity` and for
     * symbols 1 ... `alphabet`; NULL / 0 where that does not apply */
    uint64_t const *runes_needed; /* runes the buffer must hold for this batch (utf8.hip: strings beyond the capacity were skipped) */
    uint64_t runes_capacity;
    uint32_t const *alphabet_flags; /* [any_multibyte, distinct runes, table overflowed] of szs_hip_alphabet_rename */
    uint32_t alphabet;              /* > 0: the kernels index a direct table of alphabet + 1 rows with the symbols */
} szs_plan_expectation_t;

#define SZS_PLAN_RANK_SAMPLES 32u /* the length distribution of a side, for the queue order of hip/myers_queue.hip (host/plan.c) */
typedef struct szs_plan_summary_t {
    uint32_t status;           /* SZS_PLAN_STATUS_* bits; 0 = both sides planned */
    uint32_t speculation_held; /* the batch fits the expectation: the speculated launches scored it */
    szs_side_stats_t side[2];  /* queries, candidates (symmetric: twice the same) */
    uint32_t variant_counts[2][SZS_PLAN_VARIANTS];
    uint64_t symmetric_cells;  /* symmetric calls: cells of the lower triangle */
    uint32_t sequence;
    /* rank_lengths[side][k]: the length of the string at ASCENDING rank k (count - 1) / SZS_PLAN_RANK_SAMPLES of that side -
     * [0] the shortest, [SZS_PLAN_RANK_SAMPLES] the longest; every string of a lower rank is no longer than the sample */
    uint32_t rank_lengths[2][SZS_PLAN_RANK_SAMPLES + 1];
} szs_plan_summary_t;

/**
 *  Plans one call on the device: reads the offsets of both tapes (`candidates` NULL: symmetric), writes for each side the
 *  refs sorted by ascending and by descending length, and the summary (pinned host memory is fine).  With `expected`
 *  enabled and violated, the refs of the side that violates it are written with length 0.  One workgroup per side; see
 *  hip/planner.hip.
 */
int szs_hip_plan(szs_plan_side_t const *queries, szs_plan_side_t const *candidates, unsigned myers_words,
                 szs_plan_expectation_t const *expected, szs_plan_summary_t *summary,
                 uint32_t *verdicts /* 8 dwords of device memory, zeroed when allocated and never again: where the two workgroups of a
                                       two-sided plan leave their verdicts and count themselves (NULL allowed when `candidates` is) */,
                 void *stream);

/**
 *  The planner FOLDED INTO the scoring launch (hip/lev_myers.hip, round 5): a unit-cost byte call whose queries all fit the
 *  short kernel (<= 256 bytes) and whose sides hold at most SZS_FUSED_MOST_STRINGS strings is ONE launch and nothing else.
 *  Workgroup 0 sorts the kernel's query side, workgroup 1 its candidate side (a counting sort of the lengths in LDS, 256
 *  threads, ~2 us), each writes the side's ascending and descending refs - exactly what szs_hip_plan writes - and publishes
 *  `ready[side] = sequence` (release, agent scope); every workgroup waits for both words (acquire) before it reads a ref.
 *  Workgroups are dispatched in order, so the two sorters are resident before anyone can wait for them.  No planner launch,
 *  no kernel boundary between planning and scoring (config 2: 12 + 5 of a 202 us call).
 *  A side whose offsets are malformed, or a query side with a string beyond 256 bytes, is written BLANK (every length 0:
 *  the launch scores empty strings, memory-safe) and says so in its report; the host then plans the call the ordinary way.
 */
#define SZS_FUSED_MOST_STRINGS 1024u /* per side, sorted in ONE pass: four strings per thread of the sorting workgroup, all in registers
                                        (eight cost the scoring bodies a wavefront of occupancy: 108 VGPRs against 95) */
#define SZS_FUSED_MOST_STRINGS_TWO_PASSES 16384u /* per side, round 6: larger sides are counted in one walk over their offsets and
                                        placed in a second (the offsets come from the L2 the second time; the refs go straight to memory) */
#define SZS_FUSED_BINS 1024u         /* lengths of 1023 bytes and more share the last bin (they are texts: any order scores the same) */
typedef struct szs_fused_side_report_t {
    uint32_t sequence; /* of the launch that wrote this report */
    uint32_t status;   /* SZS_PLAN_STATUS_DESCENDING | SZS_PLAN_STATUS_OVERFLOW */
    uint32_t blank;    /* the side's refs were written with length 0: nothing real was scored */
    uint32_t reserved;
    szs_side_stats_t stats;
    uint32_t rank_lengths[SZS_PLAN_RANK_SAMPLES + 1];
    uint32_t ticks[5]; /* 100 MHz: the sorter's begin; offsets loaded; positions known; refs written; published - all relative to [0] but [0] itself */
    uint32_t padding;
    uint64_t squares;  /* sum of the squared lengths: a symmetric call's cells are ((sum of lengths)^2 + squares) / 2 */
} szs_fused_side_report_t;
typedef struct szs_fused_plan_t {
    szs_plan_side_t side[2]; /* KERNEL roles: [0] its queries (patterns; scored from .descending), [1] its candidates (.ascending) */
    uint32_t sequence;       /* never 0 */
    uint32_t *ready;         /* device memory: ready[0] and ready[32], zeroed when allocated; a launch leaves `sequence` in both */
    szs_fused_side_report_t *report; /* [2], pinned host memory */
    uint32_t *gave_up;       /* pinned host memory: a waiting workgroup whose polls ran out leaves `sequence` here and scores nothing -
                                the host then plans the call the ordinary way (a launch never hangs on a sorter that does not publish) */
    uint32_t poll_budget;    /* polls of a `ready` word a waiting workgroup spends before it gives up (~1 us each) */
    uint32_t withhold;       /* testing aid (`fused` knob = 2): the sorters sort and report but never publish */
    uint32_t symmetric;      /* one side only: side[0] is sorted by workgroup 0 and serves both roles; report[1] is not written */
    uint32_t blocks_per_group; /* candidate blocks a workgroup walks (1: the plain grid; the launcher merges blocks of tiny strings) */
} szs_fused_plan_t;
#define SZS_FUSED_POLL_BUDGET (1u << 18) /* ~0.2 s: four orders of magnitude beyond the ~4 us a sorter takes */
int szs_hip_levenshtein_myers_fused(szs_fused_plan_t const *plan, uint64_t *results, uint64_t results_row_stride, int layout,
                                    void *stream);

/**
 *  Transcodes `count` UTF-8 strings (byte refs) into UTF-32 with the value contract of `sz_rune_decode_unchecked`:
 *  string i's runes land at `runes + rune_starts[i]`, its rune count in `rune_counts[i]`; `*any_multibyte` is OR-ed
 *  with 1 when any string holds a byte >= 0x80 (the caller zeroes it).  One thread per string.
 */
int szs_hip_utf8_transcode(szs_string_ref_t const *strings, uint32_t count, uint64_t const *rune_starts, uint32_t *runes,
                           uint32_t *rune_counts, uint32_t *any_multibyte, void *stream);

/**
 *  The same from a TAPE whose offsets the host has not read (the device-planned codepoint path): string i's runes start at
 *  side_base + align4(offsets[i] - offsets[0]) + 8 i, side_base = 0 or the span of the `before` tape; the starts are written to
 *  `rune_starts` (for szs_hip_alphabet_rename and the planner).  Strings whose slot would pass `capacity` runes are skipped
 *  and `*needed` (device memory) receives the runes the buffer must hold: the caller compares, grows and repeats.
 */
int szs_hip_utf8_transcode_tape(void const *data, void const *offsets, uint32_t count, int wide, void const *before_offsets,
                                uint32_t before_count, int before_wide, uint64_t capacity, uint32_t *runes, uint64_t *rune_starts,
                                uint32_t *rune_counts, uint32_t *any_multibyte, uint64_t *needed, void *alphabet_workspace, void *stream);
/** Both tapes of a call in one launch: the second tape's runes follow the first's (`rune_starts` / `rune_counts`: first_count
 *  entries, then second_count); second_count 0: the first tape alone.  `alphabet_workspace` (or NULL): the table of the
 *  szs_hip_alphabet_rename that follows, emptied by this launch instead of by two fills of its own (`workspace_is_empty` there). */
int szs_hip_utf8_transcode_tapes(void const *first_data, void const *first_offsets, uint32_t first_count, int first_wide,
                                 void const *second_data, void const *second_offsets, uint32_t second_count, int second_wide,
                                 uint64_t capacity, uint32_t *runes, uint64_t *rune_starts, uint32_t *rune_counts,
                                 uint32_t *any_multibyte, uint64_t *needed, void *alphabet_workspace, void *stream);

/**
 *  Renumbers the runes of a transcoded batch 1 ... A (equal runes, equal ids) in place, when it holds at most `most` distinct
 *  ones; `alphabet_out[0]` receives the number of distinct runes, `alphabet_out[1]` 1 when the table overflowed (device
 *  memory; the arrays are renamed iff `alphabet_out[0] <= most && !alphabet_out[1]`).  Does nothing when `*any_multibyte` is 0.
 *  The codepoint kernels take the alphabet's size and look symbols up in a direct table instead of probing a hash table.
 */
#define SZS_ALPHABET_SLOTS (1u << 16)
#define SZS_ALPHABET_MOST 4095u
#define SZS_ALPHABET_WORTH_BYTES (1u << 16) /* smaller batches keep their runes: three launches cost more than the probes */
int szs_hip_alphabet_rename(uint32_t count, uint64_t const *rune_starts, uint32_t const *rune_counts, uint32_t *runes,
                            uint32_t const *any_multibyte, void *workspace, int workspace_is_empty, uint32_t most, uint32_t *alphabet_out,
                            void *stream);
size_t szs_hip_alphabet_workspace_bytes(void);

/** Scoring model handed to the weighted kernels; lives in device memory, one per engine. */
typedef struct szs_cost_model_t {
    int16_t substitution[32 * 32]; /* [query class][candidate class]; Levenshtein engines: negated costs */
    uint8_t byte_to_class[256];
    int32_t gap_open;   /* signed, ADDED (negated for Levenshtein) */
    int32_t gap_extend; /* == gap_open for linear gaps */
    int32_t uniform_match, uniform_mismatch; /* Levenshtein engines: negated uniform costs; unused otherwise */
} szs_cost_model_t;

enum {
    szs_objective_global_k = 0,      /* Needleman-Wunsch: bottom-right cell */
    szs_objective_local_k = 1,       /* Smith-Waterman: best cell, substitution branch clamped at 0 */
    szs_objective_distance_k = 2,    /* weighted Levenshtein: global on negated uniform costs, result negated */
    szs_objective_distance_runes_k = 3, /* the same over UTF-32 strings (addresses point at u32 runes, lengths in runes) */
    szs_objective_local_saturating_k = 4 /* Smith-Waterman with gap costs <= 0: unsigned-saturating gap arithmetic */
};

/**
 *  Weighted scorer: one (query, candidate) pair per lane, the query shared by the wavefront, DP walked in strips
 *  of query rows held in registers, the strip boundary row parked in `boundary` (global memory, [column][lane]).
 *  `boundary` needs szs_hip_weighted_boundary_bytes(...) bytes (work counter + one boundary per RESIDENT workgroup)
 *  and must be called with the target device current.  Queries should arrive longest first.
 *  `narrow` != 0 promises that every value parked on a strip boundary fits int16 (global objectives: the reach of
 *  serial.hpp:135-162; saturating local: shortest side x largest cost); boundaries are then stored in 16 bits.
 */
int szs_hip_weighted_scores(int objective, int affine, int narrow, szs_cost_model_t const *model, szs_string_ref_t const *queries,
                            uint32_t queries_count, szs_string_ref_t const *candidates, uint32_t candidates_count,
                            uint32_t longest_candidate, int64_t *results, uint64_t results_row_stride, int symmetric,
                            void *boundary, void *stream);
size_t szs_hip_weighted_boundary_bytes(int objective, int affine, int narrow, uint32_t queries_count, uint32_t candidates_count,
                                       uint32_t longest_candidate);

/**
 *  Two cells per VALU operation (hip/weighted_packed.hip): the same lanes-tier scorer for class-table engines whose DP
 *  values provably fit int16 - `local` = 0: Needleman-Wunsch with reach (serial.hpp:135-162) < 32000; `local` = 1:
 *  Smith-Waterman with both gap costs <= 0 and (shortest side + 3) x largest cost < 32000.  `classes` = 1 + the largest
 *  value of `byte_to_class` (<= 32).  Workspace and calling rules as for szs_hip_weighted_scores.
 */
int szs_hip_weighted_packed_scores(int local, int affine, uint32_t classes, szs_cost_model_t const *model,
                                   szs_string_ref_t const *queries, uint32_t queries_count, szs_string_ref_t const *candidates,
                                   uint32_t candidates_count, uint32_t longest_candidate, int64_t *results,
                                   uint64_t results_row_stride, int symmetric, void *boundary, void *stream);
size_t szs_hip_weighted_packed_boundary_bytes(int local, int affine, uint32_t classes, uint32_t queries_count,
                                              uint32_t candidates_count, uint32_t longest_candidate);

/**
 *  The TEAM tier of the same 16-bit scorers (hip/weighted_teams.hip): a (pair of queries, candidate) item is scored by a team
 *  of `lanes` adjacent lanes, each `registers` query rows deep, strips handed from lane to lane in registers and only the
 *  bottom row of a whole group of lanes x registers rows parked; two queries share every register (low / high half).
 *  Same eligibility, cost model, refs (queries longest first, candidates ascending) and result addressing as
 *  szs_hip_weighted_packed_scores.  `shape` = lanes * 10000 + registers * 100 + wavefronts per SIMD names one of the
 *  compiled instances: szs_hip_weighted_team_shape(i) enumerates them (0 past the last one).  `wide` = 0: cells ordered as
 *  half-float patterns, three-input maxima; = 1: as unsigned integers, two-input maxima.  The caller's bound on every DP value
 *  (global / distance: the reach of serial.hpp:135-162; local: (shorter side + 3) x largest cost) must stay below
 *  szs_hip_weighted_team_reach_limit(objective, wide): 15000 / 29000 / 30000 narrow, 32000 / 62000 / 64000 wide.
 *  `objective`: 0 Needleman-Wunsch, 1 Smith-Waterman with gap costs <= 0, 2 Levenshtein with uniform costs - the model then
 *  holds the NEGATED costs, `byte_to_class` the dense alphabet of the batch (up to 256 classes; szs_hip_byte_presence), and
 *  results are distances.
 */
unsigned szs_hip_weighted_team_shape(unsigned index);
int szs_hip_weighted_team_has_shape(unsigned shape);
uint32_t szs_hip_weighted_team_reach_limit(int objective, int wide);
int szs_hip_weighted_team_fits(unsigned shape, uint32_t classes);
/** Candidates per work item of a compiled shape (= teams per workgroup: 256 / lanes, or 512 / lanes for teams of more than sixteen lanes); 0: no such shape. */
unsigned szs_hip_weighted_team_candidates_per_item(unsigned shape);
/** OR-s into `presence[8]` (device memory, zeroed by the caller) one bit per byte value that occurs in the tape's strings. */
int szs_hip_byte_presence(void const *data, void const *offsets, uint32_t count, int wide, uint32_t *presence, void *stream);
int szs_hip_weighted_team_scores(int objective, int affine, int wide, unsigned shape, uint32_t classes, szs_cost_model_t const *model,
                                 szs_string_ref_t const *queries, uint32_t queries_count, szs_string_ref_t const *candidates,
                                 uint32_t candidates_count, uint32_t longest_candidate, int64_t *results,
                                 uint64_t results_row_stride, int symmetric, void *workspace, void *stream);
size_t szs_hip_weighted_team_workspace_bytes(int objective, int affine, int wide, unsigned shape, uint32_t classes, uint32_t queries_count,
                                             uint32_t candidates_count, uint32_t longest_candidate);

/**
 *  The 64-bit cell tier (hip/wide.hip): every objective above on an anti-diagonal walker with int64 cells, one pair per
 *  workgroup - for inputs whose reach (serial.hpp:135-162) leaves 32 bits, where the reference widens its cells too
 *  (serial.hpp:370-386, cuda.cuh:5863-5874).  Same refs, cost model and result addressing as szs_hip_weighted_scores;
 *  `workspace` holds szs_hip_wide_workspace_bytes(...) bytes (a work counter + 3 or 7 diagonals per resident pair).
 */
int szs_hip_wide_scores(int objective, int affine, szs_cost_model_t const *model, szs_string_ref_t const *queries,
                        uint32_t queries_count, szs_string_ref_t const *candidates, uint32_t candidates_count,
                        uint32_t longest_query, uint32_t longest_candidate, int64_t *results, uint64_t results_row_stride,
                        int layout, void *workspace, void *stream);
size_t szs_hip_wide_workspace_bytes(int affine, uint32_t queries_count, uint32_t candidates_count, uint32_t longest_query,
                                    uint32_t longest_candidate);

/**
 *  The few-pairs tier of the weighted scorers (hip/systolic.hip): a pair is spread over wavefronts - 64 lanes x R rows
 *  per band, lanes skewed by one column and chained by DPP, bands chained through memory - instead of owning one lane.
 *  Same objectives, cost model, string refs and result addressing as szs_hip_weighted_scores; queries need no
 *  particular order.  Two device blocks, sized by szs_hip_systolic_workspace_bytes (returns 0 when the job is too
 *  large for this tier):
 *    `control` - epoch-tagged 64-bit words (ticket counter, stall flag, per-band progress, per-pair best / done).  The
 *                caller zeroes the block ONCE when it allocates it and passes a strictly increasing `epoch` >= 1 with
 *                every launch that uses it; nothing is ever cleared between launches.
 *    `parked`  - the band bottom rows in flight; contents never matter.
 *  Both must be called with the target device current.  After the launch has completed, the 64-bit word at
 *  `control + 8` equals ((uint64_t)epoch << 32 | 1) if a band gave up waiting for its predecessor (a broken invariant,
 *  never expected): the results are then invalid and the host reports the failure instead of hanging the device.
 */
#define SZS_SYSTOLIC_BAND_ROWS 512u /* 64 lanes x 8 rows; szs_hip_systolic_band_rows() returns the same */
unsigned szs_hip_systolic_band_rows(void);
int szs_hip_systolic_workspace_bytes(int affine, uint32_t queries_count, uint32_t candidates_count, uint32_t longest_query,
                                     uint32_t longest_candidate, size_t *control_bytes, size_t *parked_bytes);
int szs_hip_systolic_scores(int objective, int affine, szs_cost_model_t const *model, szs_string_ref_t const *queries,
                            uint32_t queries_count, szs_string_ref_t const *candidates, uint32_t candidates_count,
                            uint32_t longest_query, uint32_t longest_candidate, int64_t *results,
                            uint64_t results_row_stride, int symmetric, void *control, void *parked, uint32_t epoch,
                            void *stream);

/**
 *  Unit-cost byte-level Levenshtein for few, long pairs (hip/myers_chain.hip): the bit-parallel recurrence on the band
 *  chain of the systolic tier - 64 lanes x 32 rows per band, 8 columns per step, one dword of horizontal deltas handed
 *  from lane to lane and from band to band.  Any query length.  Control block and epoch exactly as for
 *  szs_hip_systolic_scores (the two kernels may share one control block); `parked` holds one dword per step and pair.
 */
#define SZS_MYERS_CHAIN_BAND_ROWS 2048u
int szs_hip_myers_chain_workspace_bytes(uint32_t queries_count, uint32_t candidates_count, uint32_t longest_query,
                                        uint32_t longest_candidate, size_t *control_bytes, size_t *parked_bytes);
int szs_hip_myers_chain(szs_string_ref_t const *queries, uint32_t queries_count, szs_string_ref_t const *candidates,
                        uint32_t candidates_count, uint32_t longest_query, uint32_t longest_candidate, uint64_t *results,
                        uint64_t results_row_stride, int layout_flags, void *control, void *parked, uint32_t epoch,
                        void *stream);

/**
 *  Rolling MinHash / Count-Min fingerprints (hip/fingerprints.hip).  One lane per dimension, one workgroup per 256
 *  dimensions x one SEGMENT (4096 window positions) of one text; texts of several segments are merged by a second kernel.
 *  Host-prepared tables, all in device memory: `segment_text[segment]` (NULL when every text has exactly one segment),
 *  `segment_prefix[text]` / `partial_prefix[text]` (first segment / first partial slot of each text; count + 1 entries),
 *  `merge_list[merge_count]` (slots of the texts with more than one segment), the per-dimension parameters, and
 *  `partial_*` with one (double, u32) per (segment of a multi-segment text, dimension).  Output strides in BYTES.
 */
#define SZS_FINGERPRINT_SEGMENT 4096u
#define SZS_FINGERPRINT_MAX_WIDTH 1024u   /* windows up to here find their bytes staged in LDS */
#define SZS_FINGERPRINT_WIDEST 65536u    /* wider ones (round 4) read the text where it lies; the engine takes [2, this] */
int szs_hip_fingerprints(szs_string_ref_t const *texts, uint32_t texts_count, uint32_t const *segment_text,
                         uint32_t const *segment_prefix, uint32_t const *partial_prefix, uint32_t total_segments,
                         uint32_t const *merge_list, uint32_t merge_count, uint32_t dimensions, uint32_t const *widths,
                         double const *multipliers, double const *modulos, double const *reciprocals,
                         double const *complements, double *partial_minimums, uint32_t *partial_counts,
                         uint32_t *min_hashes, uint64_t min_hashes_stride, uint32_t *min_counts,
                         uint64_t min_counts_stride, uint32_t widest /* the engine's widest window */, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* SZS_ROCM_KERNELS_H_ */
