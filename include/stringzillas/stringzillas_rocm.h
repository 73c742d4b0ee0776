/*
 *  stringzillas_rocm.h - ADDITIVE entry points of the ROCm build.  Nothing here changes a reference signature;
 *  a caller that only knows <stringzillas/stringzillas.h> never needs this header.
 *
 *  - szs_rocm_last_call_profile : device-event kernel time and work counters of the last engine call - the
 *    counterpart of the reference's `cuda_status_t::elapsed_milliseconds` / "Kernel GCUPS"
 *    (/root/reference/include/stringzillas/types.cuh:280-298,482-534; bench/similarities.cuh:303-308).
 *  - szs_rocm_shard_rows        : longest-processing-time assignment of query rows to N GPUs;
 *    the reference has no multi-GPU path at all (one engine call = one device, stringzillas.h:137).
 *  - szs_rocm_plan_probe, szs_rocm_orientation_probe : expose the host planner so it can be unit-tested without a GPU.
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
    sz_u32_t planner;             /* 0: planned on the host; 1: on the device (hip/planner.hip); 2: on the device, launches speculated */
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
 *  Tuning / testing knobs (csrc/host/tuning.c).  The library reads the `SZS_ROCM_*` environment variables ONCE, when it
 *  is loaded; afterwards a knob changes only through this call.  `knob` is one of "tier" (lanes | systolic | chain),
 *  "swap" (0 | 1), "packed" (0), "rune_ids" (n), "chain_waves" (4 | 8 | 16), "trace" (0 | 1), "cells" (64),
 *  "planner" (host | device), "speculate" (0) - or its environment spelling ("SZS_ROCM_TIER" ...); `value` NULL, "" or
 *  "auto" restores the automatic choice.  No knob changes a result: they pick among kernels that compute the same scores.
 */
SZ_API_RUNTIME sz_status_t szs_rocm_tuning_set(char const *knob, char const *value);

#ifdef __cplusplus
}
#endif
#endif /* STRINGZILLAS_ROCM_H_ */
