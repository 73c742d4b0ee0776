/*
 *  stringzillas_rocm.h - ADDITIVE entry points of the ROCm build.  Nothing here changes a reference signature;
 *  a caller that only knows <stringzillas/stringzillas.h> never needs this header.
 *
 *  - szs_rocm_last_call_profile : device-event kernel time and work counters of the last engine call - the
 *    counterpart of the reference's `cuda_status_t::elapsed_milliseconds` / "Kernel GCUPS"
 *    (/root/reference/include/stringzillas/types.cuh:280-298,482-534; bench/similarities.cuh:303-308).
 *  - szs_rocm_shard_rows        : longest-processing-time assignment of query rows to N GPUs;
 *    the reference has no multi-GPU path at all (one engine call = one device, stringzillas.h:137).
 *  - szs_rocm_plan_probe, szs_rocm_orientation_probe, szs_rocm_team_orientation_probe, szs_rocm_launch_order_probe,
 *    szs_rocm_queue_probe : expose
 *    the host planner - refs, tier and orientation, lanes per item, launch shapes and order - so that it is unit-tested
 *    without a GPU (tests/test_host_logic.py).
 *  - szs_rocm_node_*            : one cross-product over the N GPUs of a host, in C (csrc/host/node.c).
 *  - szs_rocm_tuning_set        : the tuning / testing knobs.
 */
#ifndef STRINGZILLAS_ROCM_H_
#define STRINGZILLAS_ROCM_H_

#include "stringzillas.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct szs_rocm_call_profile_t {
    double kernel_milliseconds;   /* hipEvent pair around the scoring launches, on the scope's stream */
    double host_milliseconds;     /* wall time of the whole C-ABI call, planning and copies included */
    sz_u64_t cells;               /* sum over scored pairs of len(query) * len(candidate): the GCUPS numerator */
    sz_u64_t pairs;               /* scored pairs (lower triangle only in symmetric mode) */
    sz_u64_t algorithmic_bytes;   /* sum over pairs of len(q) + len(c) + 2 * 4 + 8   */
    sz_u64_t unique_bytes;        /* bytes of both tapes + offsets + the results matrix, each counted once */
    sz_u32_t launches;            /* kernel launches issued */
    sz_u32_t longest_query;
    sz_u32_t longest_candidate;
    sz_u32_t tier;                /* 0: one pair per lane (lev_myers.hip, weighted*.hip); 1: systolic.hip; 2: myers_chain.hip */
    sz_u32_t transposed;          /* 1: the planner swapped the sides (candidates on workgroups, queries on lanes) */
    sz_u32_t cell_bits;           /* width of the DP cells of the last launch: 16 (weighted_packed.hip), 32, 64 (wide.hip), or 0 (bit-parallel) */
    sz_u32_t planner;             /* 0: planned on the host; 1: on the device (hip/planner.hip); 2: on the device, launches speculated;
                                     3: the plan of the previous call of the same tapes, re-used behind a guard;
                                     4: planned INSIDE the scoring launch (its first two workgroups sort the sides; hip/lev_myers.hip);
                                     5: not planned at all - the tiny-token kernel scores straight from the tapes (hip/myers_tiny.hip) */
    sz_u32_t team;                /* 0, or lanes * 10000 + registers * 100 + wavefronts per SIMD of the team tier (weighted_teams.hip) */
    sz_u32_t team_wide;           /* team tier: 0 cells ordered as half-float patterns (three-input maxima), 1 as unsigned integers */
    sz_u32_t streams;             /* streams the launches of the call were dealt over: 1 ... 8, never more than the `queues` knob */
    sz_u32_t queue_items;         /* 0, or the work items of the ONE persistent launch that scored every bit-parallel width of the call (myers_queue.hip) */
    sz_u32_t queue_tiles;         /* ... and the (query slice) x (candidate column) tiles its queue was ordered by */
} szs_rocm_call_profile_t;

/** Copies the profile of the most recent call made through `engine` (any of the four engine handle types). */
SZ_API_RUNTIME sz_status_t szs_rocm_last_call_profile(void *engine, szs_rocm_call_profile_t *profile);

/**
 *  Deals `rows` query rows to `shards` devices so that the summed `row_weights` per shard are as equal as the
 *  longest-processing-time heuristic makes them (sort descending, always give to the lightest shard).
 *  `shard_of_row[i]` receives the shard of row i; `shard_loads` (optional) the resulting per-shard weight sums.
 */
SZ_API_RUNTIME sz_status_t szs_rocm_shard_rows(sz_size_t const *row_weights, sz_size_t rows, sz_size_t shards,
                                               sz_u32_t *shard_of_row, sz_u64_t *shard_loads);

/**
 *  Deals the rows of a SYMMETRIC call's lower triangle to `shards` devices as contiguous bands of equal weight - row i weighs
 *  (len_i + 1) x sum_{j <= i} (len_j + 1), its cells - band g = rows [band_first[g], band_first[g + 1]);
 *  `band_first` holds shards + 1 entries, `band_weights` (optional) the weights dealt.  A band is its rows against every string
 *  before it (a rectangle) plus the triangle of its own rows: two ordinary calls of a single-GPU engine.
 */
SZ_API_RUNTIME sz_status_t szs_rocm_shard_triangle(sz_size_t const *lengths, sz_size_t rows, sz_size_t shards, sz_size_t *band_first,
                                                   sz_u64_t *band_weights);

/**
 *  Runs the host planner on bare length arrays.  Outputs (all optional):
 *    candidate_order[c]  - candidate indices by ascending length (stable);
 *    query_order[q]      - query indices grouped by kernel variant;
 *    query_variant[q]    - the variant (Myers: 32-bit words rounded to an instantiated kernel; 0 = weighted kernel)
 *                          of the query at planned position q;
 *    cells               - sum of len(q) * len(c) over live pairs.
 */
SZ_API_RUNTIME sz_status_t szs_rocm_plan_probe(int unit_cost, int symmetric, sz_u32_t const *query_lengths,
                                               sz_size_t queries_count, sz_u32_t const *candidate_lengths,
                                               sz_size_t candidates_count, sz_u32_t *candidate_order,
                                               sz_u32_t *query_order, sz_u32_t *query_variant, sz_u64_t *cells);

/**
 *  Runs the planner's tier / orientation decision on bare length arrays (no GPU needed): `*tier` receives 0 (one pair per
 *  lane) or 1 (systolic: one pair per chain of wavefronts), `*transposed` whether the sides are swapped so that the
 *  candidates take the workgroup / band role.  `unit_cost`: the engine is unit-cost Levenshtein (bit-parallel kernels
 *  exist); `uniform`: any Levenshtein engine; `candidate_lengths` is ignored for symmetric calls.
 */
SZ_API_RUNTIME sz_status_t szs_rocm_orientation_probe(int unit_cost, int affine, int uniform, int symmetric,
                                                      sz_u32_t const *query_lengths, sz_size_t queries_count,
                                                      sz_u32_t const *candidate_lengths, sz_size_t candidates_count,
                                                      int *tier, int *transposed);

/**
 *  The same decision for a class-table engine whose DP values fit 16 bits - the calls the team tier (hip/weighted_teams.hip)
 *  may take: `*lanes` receives the lanes per (pair of queries, candidate) the planner deals, 16 or 4, or 0 for the
 *  one-pair-per-lane kernel (also when `*tier` is not 0).
 */
SZ_API_RUNTIME sz_status_t szs_rocm_team_orientation_probe(int affine, int symmetric, sz_u32_t const *query_lengths,
                                                           sz_size_t queries_count, sz_u32_t const *candidate_lengths,
                                                           sz_size_t candidates_count, int *tier, int *transposed, sz_u32_t *lanes);

/**
 *  The launches of a unit-cost Levenshtein call over queries of these lengths (bytes, or runes with `runes` != 0) against
 *  `candidates_count` candidates, in the order they leave the host - longest pair first, the short launch last
 *  (DESIGN.md section 4.1): per launch the width group's variant (8: the short kernel; 10 ... 64 words; 0: longer queries),
 *  the words of the kernel that takes it and the lanes per pair (0: one).  `*launches` receives the number of groups; at
 *  most `capacity` entries are written.  No GPU involved.
 */
SZ_API_RUNTIME sz_status_t szs_rocm_launch_order_probe(int runes, sz_u32_t const *query_lengths, sz_size_t queries_count,
                                                       sz_size_t candidates_count, sz_u32_t *variants, sz_u32_t *words, sz_u32_t *lanes,
                                                       sz_size_t capacity, sz_size_t *launches);

/**
 *  The work queue of the ONE persistent launch that scores every bit-vector width of a unit-cost byte call
 *  (hip/myers_queue.hip; host/plan.c: szs_plan_queue), planned from bare length arrays - no GPU involved.  Queries are taken
 *  longest first, candidates shortest first (szs_rocm_plan_probe gives both orders).  `alphabet` 0: byte strings; A: lengths
 *  count codepoints of a batch renumbered 1 ... A (hip/utf8.hip), whose tables have A + 1 rows - `*items_total` 0 when such an
 *  alphabet leaves some query no table (the per-width launches score those calls).  `tiles` receives 10 values per tile, in
 *  queue order: items of all tiles before it, first query and queries of its slice, first and one-past-last candidate of its
 *  column, candidates per work item, words per lane (0: the query's own width on one lane), lanes per pair, queries per
 *  work item G and flags (1: sparse tables).  With `groups` = ceil(queries / G), work item j of a tile scores the queries of group `j % groups` against the
 *  candidates of block `j / groups`, blocks cut from the column's end.
 */
SZ_API_RUNTIME sz_status_t szs_rocm_queue_probe(int symmetric, sz_u32_t alphabet, sz_u32_t const *query_lengths, sz_size_t queries_count,
                                                sz_u32_t const *candidate_lengths, sz_size_t candidates_count, sz_u32_t *tiles,
                                                sz_size_t capacity, sz_size_t *tiles_count, sz_u64_t *items_total);

/* ---- one cross-product over the N GPUs of a host (csrc/host/node.c) ------------------------------
 *
 *  The reference's C-ABI is one device per call (stringzillas.h:137) and has no multi-GPU path; what its threading rules
 *  allow - N scopes and N engines on N host threads - is packaged here: query ROWS are dealt to the GPUs by LPT on their
 *  lengths, both tapes are replicated to every GPU once per call (peer-to-peer over xGMI when they live on a GPU), one host
 *  thread per GPU runs the ordinary single-GPU engine on `its rows x all candidates`, and every result row is copied to its
 *  place in the caller's matrix.  Scores are bit-identical to the single-GPU engines': they ARE the single-GPU engines.
 */
#define SZS_ROCM_NODE_MOST_GPUS 16

typedef void *szs_rocm_node_t;        /* a set of GPUs of this host */
typedef void *szs_rocm_node_engine_t; /* one cost model, instantiated on every GPU of a node */

typedef struct szs_rocm_node_stats_t {
    sz_size_t gpus;
    double wall_milliseconds;                             /* the whole call */
    double busy_milliseconds[SZS_ROCM_NODE_MOST_GPUS];    /* per GPU: replication + scoring + placing its rows */
    double kernel_milliseconds[SZS_ROCM_NODE_MOST_GPUS];  /* per GPU: the scoring kernels alone (hipEvent pair) */
    sz_u64_t cells[SZS_ROCM_NODE_MOST_GPUS];              /* per GPU: DP cells scored */
    sz_u64_t row_weights[SZS_ROCM_NODE_MOST_GPUS];        /* per GPU: sum of (len(query) + 1) over its rows - what LPT balances */
    sz_u32_t rows[SZS_ROCM_NODE_MOST_GPUS];               /* per GPU: query rows dealt to it */
    sz_u32_t peer_copies[SZS_ROCM_NODE_MOST_GPUS];        /* per GPU: tape replicas that came straight from another GPU's memory (peer access over xGMI) */
    sz_u32_t staged_copies[SZS_ROCM_NODE_MOST_GPUS];      /* per GPU: replicas staged through pinned host memory (no peer access to the source GPU), or from host memory */
    sz_u32_t peer_pairs;                                  /* ordered pairs of the node's GPUs with peer access enabled (szs_rocm_node_init) */
    sz_u32_t symmetric;                                   /* 1: the call sharded the lower triangle (bands of rows) and mirrored it */
} szs_rocm_node_stats_t;

/** `gpu_devices` NULL or `count` 0: every visible GPU.  `*node` receives the handle. */
SZ_API_RUNTIME sz_status_t szs_rocm_node_init(sz_size_t const *gpu_devices, sz_size_t count, szs_rocm_node_t *node,
                                              char const **error_message);
SZ_API_RUNTIME sz_size_t szs_rocm_node_size(szs_rocm_node_t node);
/** Engines created from the node keep it alive until they are freed themselves.  The handle must not be used after this call:
 *  a second free is recognised (and ignored) only while such engines still exist - afterwards the memory is gone. */
SZ_API_RUNTIME void szs_rocm_node_free(szs_rocm_node_t node);

/** Engines of a node: same arguments and meaning as `szs_*_init` (stringzillas.h), `*engine` must be NULL on entry. */
SZ_API_RUNTIME sz_status_t szs_rocm_node_levenshtein_distances_init(szs_rocm_node_t node, sz_error_cost_t match,
                                                                    sz_error_cost_t mismatch, sz_error_cost_t open,
                                                                    sz_error_cost_t extend, szs_rocm_node_engine_t *engine,
                                                                    char const **error_message);
SZ_API_RUNTIME sz_status_t szs_rocm_node_levenshtein_distances_utf8_init(szs_rocm_node_t node, sz_error_cost_t match,
                                                                         sz_error_cost_t mismatch, sz_error_cost_t open,
                                                                         sz_error_cost_t extend, szs_rocm_node_engine_t *engine,
                                                                         char const **error_message);
SZ_API_RUNTIME sz_status_t szs_rocm_node_needleman_wunsch_scores_init(szs_rocm_node_t node, sz_u8_t const *byte_to_class,
                                                                      sz_error_cost_t const *class_substitution_costs,
                                                                      sz_error_cost_t open, sz_error_cost_t extend,
                                                                      szs_rocm_node_engine_t *engine, char const **error_message);
SZ_API_RUNTIME sz_status_t szs_rocm_node_smith_waterman_scores_init(szs_rocm_node_t node, sz_u8_t const *byte_to_class,
                                                                    sz_error_cost_t const *class_substitution_costs,
                                                                    sz_error_cost_t open, sz_error_cost_t extend,
                                                                    szs_rocm_node_engine_t *engine, char const **error_message);
SZ_API_RUNTIME void szs_rocm_node_engine_free(szs_rocm_node_engine_t engine);

/**
 *  Scores all `queries x candidates` into `results[q * stride + c]`, 8-byte cells.  `candidates` NULL: queries against themselves
 *  - the lower triangle is scored ONCE, in bands of rows of equal weight (szs_rocm_shard_triangle), and mirrored, as the
 *  single-GPU engines do (serial.hpp:3169-3182).  8-byte
 *  cells (`sz_size_t` distances / `sz_ssize_t` scores).  Tapes may live in host, pinned, unified or any GPU's memory;
 *  so may `results`.  Synchronous.  `stats` (optional) receives the per-GPU timing the multi-GPU configs ask to report.
 */
SZ_API_RUNTIME sz_status_t szs_rocm_node_scores_u32tape(szs_rocm_node_engine_t engine, sz_sequence_u32tape_t const *queries,
                                                        sz_sequence_u32tape_t const *candidates, void *results,
                                                        sz_size_t results_row_stride, szs_rocm_node_stats_t *stats,
                                                        char const **error_message);
SZ_API_RUNTIME sz_status_t szs_rocm_node_scores_u64tape(szs_rocm_node_engine_t engine, sz_sequence_u64tape_t const *queries,
                                                        sz_sequence_u64tape_t const *candidates, void *results,
                                                        sz_size_t results_row_stride, szs_rocm_node_stats_t *stats,
                                                        char const **error_message);

/**
 *  Tuning / testing knobs (csrc/host/tuning.c).  The library reads the `SZS_ROCM_*` environment variables ONCE, when it
 *  is loaded; afterwards a knob changes only through this call.  `knob` is one of "tier" (lanes | systolic | chain),
 *  "swap" (0 | 1), "packed" (0), "rune_ids" (n), "chain_waves" (4 | 8 | 16), "trace" (0 | 1), "cells" (64),
 *  "planner" (host | device), "speculate" (0), "streams" (0: one stream), "reuse" (0: never re-use a plan),
 *  "split" (0 | 2 | 4 | 8: lanes per pair of the bit-parallel widths of 16 words and more), "alphabet" (0 | 1: never / always renumber the runes
 *  of a codepoint batch on the device), "merge" (n: candidate blocks per workgroup of the short bit-parallel kernels),
 *  "team" (0: never | lanes * 10000 + registers * 100 + waves: that shape of the team tier of the 16-bit weighted scorers),
 *  "queue" (0: never | 1: every unit-cost byte call - the one persistent launch of hip/myers_queue.hip; automatic: calls of two or
 *  more bit-vector widths whose lengths are skewed), "queue_words" (4 | 8 | 12 | 16: the most words of a pattern one lane holds
 *  there), "queue_rounds" (n: candidates per work item in rounds of eight wavefronts), "queue_priority" (0 | 1: wave priorities by
 *  chain length inside that launch; automatic: byte calls and short codepoint calls), "fused" (0: never plan a short unit-cost
 *  call inside its own scoring launch), "tiny" (0: never | 1: every unit-cost byte call of strings up to 255 bytes of which
 *  few are beyond 16 - the tiny-token launch of hip/myers_tiny.hip | 2: the same, and blocks full of longer strings are scored there
 *  too, slowly, instead of refused (testing); automatic: batches of tiny tokens on both sides),
 *  "queues" (see below), "roctx" (1: the host phases of every call - plan, decide, enqueue, wait - as roctx ranges for a
 *  `rocprofv3 --marker-trace` timeline; the marker library is looked up at run time, never linked),
 *  "cpu_requests" (strict | gpu: serve capability
 *  masks without the GPU bit and CPU device scopes with the GPU engines on device 0 instead of refusing them) - or its
 *  environment spelling ("SZS_ROCM_TIER" ...); `value` NULL, "" or "auto" restores the automatic choice.  No knob changes a
 *  result: they pick among kernels that compute the same scores.
 *
 *  "queues" (n): hardware queues the process has.  The launches of a mixed-length batch fan out over at most that many
 *  streams; the HIP runtime gives a process GPU_MAX_HW_QUEUES of them, 4 by default, fixed when HIP initialises.  The library
 *  never writes the environment: it reads GPU_MAX_HW_QUEUES once, when it is loaded, and an application that wants the wide
 *  fan-out (a latency-bound share of a batch: 2.0 ms on twelve queues, 3.0 on four) exports GPU_MAX_HW_QUEUES=12
 *  itself before its first HIP call.  `szs_rocm_call_profile_t.streams` says what a call really used.
 */
SZ_API_RUNTIME sz_status_t szs_rocm_tuning_set(char const *knob, char const *value);

/** The `team` shapes this build holds (lanes * 10000 + registers * 100 + wavefronts per SIMD), 0 past the last one. */
SZ_API_RUNTIME sz_u32_t szs_rocm_team_shape(sz_size_t index);

#ifdef __cplusplus
}
#endif
#endif /* STRINGZILLAS_ROCM_H_ */
