/*
 *  stringzillas.h - C-ABI of the ROCm build of StringZillas' batched similarity engines (MI355X / gfx950).
 *
 *  This is the drop-in boundary.  It declares - with identical names, argument order, types and status
 *  conventions - every symbol of the reference's public header
 *      /root/reference/include/stringzillas/stringzillas.h  (v5.1.2, 41 functions, lines 36-613)
 *  so that `libstringzillas_rocm_shared.so` fills the `stringzillas_rocm_shared` slot the reference declares
 *  but leaves empty (/root/reference/CMakeLists.txt:14,819; build.rs:759-763; setup.py:863-865).
 *
 *  The header is self-contained: the handful of POD types the reference pulls from
 *  include/stringzilla/types.h are restated here (layout-identical, each with its source line) inside an
 *  `#ifndef STRINGZILLA_TYPES_H_` guard, so a translation unit that already included the reference's own
 *  types header keeps those definitions.
 *
 *  Semantics every engine call follows (reference stringzillas.h:173-177):
 *      results[query_index * results_row_stride + candidate_index] = score(queries[query_index], candidates[candidate_index])
 *      candidates == NULL  =>  symmetric self-similarity of `queries`: the lower triangle (diagonal included) is
 *                              scored as (query = i, candidate = j <= i) and mirrored into the upper one.
 *      results_row_stride is in ELEMENTS and >= the number of candidates; padding columns are never written.
 *      An empty matrix (no queries or no candidates) succeeds and writes nothing.
 *
 *  What is different in the ROCm build (see DESIGN.md and INTEGRATION.md):
 *    - Only GPU engines exist.  `*_init` needs `sz_cap_cuda_k` in `capabilities` - the bit every existing
 *      caller already uses for "a GPU engine exists" (include/stringzilla/types.h:863, bindings gate unified
 *      memory on it: python/stringzillas/stringzillas.h:200-202) - and returns `sz_missing_gpu_k` otherwise.
 *      There is no CPU fallback anywhere in this library.
 *    - Strings and tapes must be device-accessible (hipMalloc, hipMallocManaged / szs_unified_alloc, or
 *      hipHostMalloc memory), else `sz_device_memory_mismatch_k` - as in the reference (cuda.cuh:4268-4272).
 *      `results` may live anywhere; plain host memory is staged through a pinned bounce buffer.
 *    - The additive, ROCm-only entry points live in <stringzillas/stringzillas_rocm.h>.
 */
#ifndef STRINGZILLAS_H_
#define STRINGZILLAS_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#ifndef STRINGZILLA_TYPES_H_ /* the reference's include/stringzilla/types.h was not seen: restate its PODs */

typedef int8_t sz_i8_t;       /* types.h:628 */
typedef uint8_t sz_u8_t;      /* types.h:629 */
typedef uint32_t sz_u32_t;    /* types.h:633 */
typedef uint64_t sz_u64_t;    /* types.h:634 */
typedef size_t sz_size_t;     /* types.h:636 - pointer-sized unsigned */
typedef ptrdiff_t sz_ssize_t; /* types.h:637 - pointer-sized signed */
typedef char const *sz_cptr_t;        /* types.h:735 */
typedef sz_i8_t sz_error_cost_t;      /* types.h:736 - one substitution / gap cost */
typedef sz_size_t sz_sorted_idx_t;    /* types.h:748 */

/* types.h:810-833 */
typedef enum sz_status_t {
    sz_success_k = 0,
    sz_bad_alloc_k = -10,
    sz_invalid_utf8_k = -12,
    sz_contains_duplicates_k = -13,
    sz_overflow_risk_k = -14,
    sz_unexpected_dimensions_k = -15,
    sz_missing_gpu_k = -16,
    sz_device_code_mismatch_k = -17,
    sz_device_memory_mismatch_k = -18,
    sz_authentication_failed_k = -19,
    sz_status_unknown_k = -1
} sz_status_t;

/* types.h:839-890.  Only the bits this library reads or reports are spelled out; values match the reference. */
typedef enum sz_capability_t {
    sz_caps_none_k = 0,
    sz_cap_serial_k = 1,
    sz_cap_parallel_k = 1 << 2,
    sz_cap_haswell_k = 1 << 5,
    sz_cap_skylake_k = 1 << 6,
    sz_cap_icelake_k = 1 << 7,
    sz_cap_neon_k = 1 << 10,
    sz_cap_sve_k = 1 << 12,
    sz_cap_rvv_k = 1 << 20,
    sz_cap_cuda_k = 1 << 21,   /* "a GPU engine exists" - the bit the ROCm build reports and requires */
    sz_cap_kepler_k = 1 << 22, /* never reported by the ROCm build */
    sz_cap_hopper_k = 1 << 23, /* never reported by the ROCm build */
    sz_caps_cuda_k = (1 << 21) | (1 << 22) | (1 << 23),
    sz_caps_cpus_k = 0x011FFCFD, /* every CPU tier bit of types.h:881-884 */
    sz_cap_any_k = 0x7FFFFFFF
} sz_capability_t;

/* types.h:1004-1019 */
typedef void *(*sz_memory_allocate_t)(sz_size_t, void *);
typedef void (*sz_memory_free_t)(void *, sz_size_t, void *);
typedef struct sz_memory_allocator_t {
    sz_memory_allocate_t allocate;
    sz_memory_free_t free;
    void *handle;
} sz_memory_allocator_t;

/* types.h:1381-1398 - strings reached through two host callbacks */
typedef sz_cptr_t (*sz_sequence_member_start_t)(void const *, sz_sorted_idx_t);
typedef sz_size_t (*sz_sequence_member_length_t)(void const *, sz_sorted_idx_t);
typedef struct sz_sequence_t {
    void const *handle;
    sz_size_t count;
    sz_sequence_member_start_t get_start;
    sz_sequence_member_length_t get_length;
} sz_sequence_t;

#endif /* STRINGZILLA_TYPES_H_ */

#ifndef SZ_API_RUNTIME
#define SZ_API_RUNTIME extern __attribute__((visibility("default")))
#endif

/* Arrow-like tapes: `count + 1` offsets, string i spans data[offsets[i] .. offsets[i + 1]).  (stringzillas.h:77-92) */
typedef struct sz_sequence_u32tape_t {
    sz_cptr_t data;
    sz_u32_t const *offsets;
    sz_size_t count;
} sz_sequence_u32tape_t;

typedef struct sz_sequence_u64tape_t {
    sz_cptr_t data;
    sz_u64_t const *offsets;
    sz_size_t count;
} sz_sequence_u64tape_t;

/* Opaque handles, all `void *` like the reference (stringzillas.h:113,179-180,337-338,512-513). */
typedef void *szs_device_scope_t;
typedef void *szs_levenshtein_distances_t;
typedef void *szs_levenshtein_distances_utf8_t;
typedef void *szs_needleman_wunsch_scores_t;
typedef void *szs_smith_waterman_scores_t;
typedef void *szs_fingerprints_t;
typedef void *szs_fingerprints_utf8_t;

/* ---- library introspection (stringzillas.h:36-70; c/stringzillas/runtime.cuh:15-56) ------------------------------ */

SZ_API_RUNTIME int szs_version_major(void);
SZ_API_RUNTIME int szs_version_minor(void);
SZ_API_RUNTIME int szs_version_patch(void);
/** What this binary ships: `sz_cap_serial_k | sz_cap_cuda_k`. */
SZ_API_RUNTIME sz_capability_t szs_capabilities_comptime(void);
/** What this machine offers: `sz_cap_serial_k`, plus `sz_cap_cuda_k` when HIP enumerates at least one device. */
SZ_API_RUNTIME sz_capability_t szs_capabilities_runtime(void);
/** The intersection of the two, cached. */
SZ_API_RUNTIME sz_capability_t szs_capabilities(void);

/* ---- memory (stringzillas.h:99,606-613; runtime.cuh:58-69,205-222) ------------------------------------------------- */

/** Fills `alloc` with the unified allocator (`hipMallocManaged`, the HIP spelling of the reference's
 *  `cuMemAllocManaged(..., CU_MEM_ATTACH_GLOBAL)`, types.cuh:145-151). */
SZ_API_RUNTIME sz_status_t sz_memory_allocator_init_unified(sz_memory_allocator_t *alloc, char const **error_message);
SZ_API_RUNTIME void *szs_unified_alloc(sz_size_t size_bytes);
SZ_API_RUNTIME void szs_unified_free(void *ptr, sz_size_t size_bytes);

/* ---- device scopes (stringzillas.h:120-171; runtime.cuh:74-201) ---------------------------------------------------- */

/** Default scope: used with a GPU engine it lazily binds device 0 (stringzillas.cuh:303-320,355-365). */
SZ_API_RUNTIME sz_status_t szs_device_scope_init_default(szs_device_scope_t *scope, char const **error_message);
/** CPU scope: representable for API compatibility, but every engine of this build is a GPU engine, so passing it
 *  to an engine call yields `sz_device_code_mismatch_k` exactly like the reference's GPU engines (levenshtein.cuh:86). */
SZ_API_RUNTIME sz_status_t szs_device_scope_init_cpu_cores(sz_size_t cpu_cores, szs_device_scope_t *scope,
                                                           char const **error_message);
/** GPU scope: one HIP device + one non-blocking stream; `sz_missing_gpu_k` when the ordinal does not exist. */
SZ_API_RUNTIME sz_status_t szs_device_scope_init_gpu_device(sz_size_t gpu_device, szs_device_scope_t *scope,
                                                            char const **error_message);
SZ_API_RUNTIME sz_status_t szs_device_scope_get_cpu_cores(szs_device_scope_t scope, sz_size_t *cpu_cores,
                                                          char const **error_message);
SZ_API_RUNTIME sz_status_t szs_device_scope_get_gpu_device(szs_device_scope_t scope, sz_size_t *gpu_device,
                                                           char const **error_message);
SZ_API_RUNTIME sz_status_t szs_device_scope_get_capabilities(szs_device_scope_t scope, sz_capability_t *capabilities,
                                                             char const **error_message);
SZ_API_RUNTIME void szs_device_scope_free(szs_device_scope_t scope);

/* ---- Levenshtein distances over bytes (stringzillas.h:197-254; c/stringzillas/levenshtein.cuh) --------------------- */

/** `*engine` must be NULL on entry.  `open == extend` selects linear gaps (levenshtein.cuh:117); match 0 /
 *  mismatch 1 / gap 1 selects the bit-parallel kernel.  `alloc` is accepted and ignored, as in the reference. */
SZ_API_RUNTIME sz_status_t szs_levenshtein_distances_init(sz_error_cost_t match, sz_error_cost_t mismatch,
                                                          sz_error_cost_t open, sz_error_cost_t extend,
                                                          sz_memory_allocator_t const *alloc,
                                                          sz_capability_t capabilities,
                                                          szs_levenshtein_distances_t *engine,
                                                          char const **error_message);
SZ_API_RUNTIME sz_status_t szs_levenshtein_distances(szs_levenshtein_distances_t engine, szs_device_scope_t device,
                                                     sz_sequence_t const *queries, sz_sequence_t const *candidates,
                                                     sz_size_t *results, sz_size_t results_row_stride,
                                                     char const **error_message);
SZ_API_RUNTIME sz_status_t szs_levenshtein_distances_u32tape(szs_levenshtein_distances_t engine,
                                                             szs_device_scope_t device,
                                                             sz_sequence_u32tape_t const *queries,
                                                             sz_sequence_u32tape_t const *candidates,
                                                             sz_size_t *results, sz_size_t results_row_stride,
                                                             char const **error_message);
SZ_API_RUNTIME sz_status_t szs_levenshtein_distances_u64tape(szs_levenshtein_distances_t engine,
                                                             szs_device_scope_t device,
                                                             sz_sequence_u64tape_t const *queries,
                                                             sz_sequence_u64tape_t const *candidates,
                                                             sz_size_t *results, sz_size_t results_row_stride,
                                                             char const **error_message);
SZ_API_RUNTIME void szs_levenshtein_distances_free(szs_levenshtein_distances_t engine);

/* ---- Levenshtein distances over UTF-8 codepoints (stringzillas.h:271-328) ------------------------------------------ */

SZ_API_RUNTIME sz_status_t szs_levenshtein_distances_utf8_init(sz_error_cost_t match, sz_error_cost_t mismatch,
                                                               sz_error_cost_t open, sz_error_cost_t extend,
                                                               sz_memory_allocator_t const *alloc,
                                                               sz_capability_t capabilities,
                                                               szs_levenshtein_distances_utf8_t *engine,
                                                               char const **error_message);
SZ_API_RUNTIME sz_status_t szs_levenshtein_distances_utf8(szs_levenshtein_distances_utf8_t engine,
                                                          szs_device_scope_t device, sz_sequence_t const *queries,
                                                          sz_sequence_t const *candidates, sz_size_t *results,
                                                          sz_size_t results_row_stride, char const **error_message);
SZ_API_RUNTIME sz_status_t szs_levenshtein_distances_utf8_u32tape(szs_levenshtein_distances_utf8_t engine,
                                                                  szs_device_scope_t device,
                                                                  sz_sequence_u32tape_t const *queries,
                                                                  sz_sequence_u32tape_t const *candidates,
                                                                  sz_size_t *results, sz_size_t results_row_stride,
                                                                  char const **error_message);
SZ_API_RUNTIME sz_status_t szs_levenshtein_distances_utf8_u64tape(szs_levenshtein_distances_utf8_t engine,
                                                                  szs_device_scope_t device,
                                                                  sz_sequence_u64tape_t const *queries,
                                                                  sz_sequence_u64tape_t const *candidates,
                                                                  sz_size_t *results, sz_size_t results_row_stride,
                                                                  char const **error_message);
SZ_API_RUNTIME void szs_levenshtein_distances_utf8_free(szs_levenshtein_distances_utf8_t engine);

/* ---- Needleman-Wunsch global scores (stringzillas.h:355-413; c/stringzillas/needleman_wunsch.cuh) ------------------ */

/** `byte_to_class[256]` and the row-major `class_substitution_costs[32 * 32]` are copied (needleman_wunsch.cuh:99-118).
 *  cost(q, c) = table[class(q)][class(c)] - query class picks the ROW (serial.hpp:199-204, parity trap for
 *  asymmetric tables).  Gap costs are signed and ADDED: pass negative penalties (serial.hpp:846-848). */
SZ_API_RUNTIME sz_status_t szs_needleman_wunsch_scores_init(sz_u8_t const *byte_to_class,
                                                            sz_error_cost_t const *class_substitution_costs,
                                                            sz_error_cost_t open, sz_error_cost_t extend,
                                                            sz_memory_allocator_t const *alloc,
                                                            sz_capability_t capabilities,
                                                            szs_needleman_wunsch_scores_t *engine,
                                                            char const **error_message);
SZ_API_RUNTIME sz_status_t szs_needleman_wunsch_scores(szs_needleman_wunsch_scores_t engine, szs_device_scope_t device,
                                                       sz_sequence_t const *queries, sz_sequence_t const *candidates,
                                                       sz_ssize_t *results, sz_size_t results_row_stride,
                                                       char const **error_message);
SZ_API_RUNTIME sz_status_t szs_needleman_wunsch_scores_u32tape(szs_needleman_wunsch_scores_t engine,
                                                               szs_device_scope_t device,
                                                               sz_sequence_u32tape_t const *queries,
                                                               sz_sequence_u32tape_t const *candidates,
                                                               sz_ssize_t *results, sz_size_t results_row_stride,
                                                               char const **error_message);
SZ_API_RUNTIME sz_status_t szs_needleman_wunsch_scores_u64tape(szs_needleman_wunsch_scores_t engine,
                                                               szs_device_scope_t device,
                                                               sz_sequence_u64tape_t const *queries,
                                                               sz_sequence_u64tape_t const *candidates,
                                                               sz_ssize_t *results, sz_size_t results_row_stride,
                                                               char const **error_message);
SZ_API_RUNTIME void szs_needleman_wunsch_scores_free(szs_needleman_wunsch_scores_t engine);

/* ---- Smith-Waterman local scores (stringzillas.h:430-488; c/stringzillas/smith_waterman.cuh) ----------------------- */

SZ_API_RUNTIME sz_status_t szs_smith_waterman_scores_init(sz_u8_t const *byte_to_class,
                                                          sz_error_cost_t const *class_substitution_costs,
                                                          sz_error_cost_t open, sz_error_cost_t extend,
                                                          sz_memory_allocator_t const *alloc,
                                                          sz_capability_t capabilities,
                                                          szs_smith_waterman_scores_t *engine,
                                                          char const **error_message);
SZ_API_RUNTIME sz_status_t szs_smith_waterman_scores(szs_smith_waterman_scores_t engine, szs_device_scope_t device,
                                                     sz_sequence_t const *queries, sz_sequence_t const *candidates,
                                                     sz_ssize_t *results, sz_size_t results_row_stride,
                                                     char const **error_message);
SZ_API_RUNTIME sz_status_t szs_smith_waterman_scores_u32tape(szs_smith_waterman_scores_t engine,
                                                             szs_device_scope_t device,
                                                             sz_sequence_u32tape_t const *queries,
                                                             sz_sequence_u32tape_t const *candidates,
                                                             sz_ssize_t *results, sz_size_t results_row_stride,
                                                             char const **error_message);
SZ_API_RUNTIME sz_status_t szs_smith_waterman_scores_u64tape(szs_smith_waterman_scores_t engine,
                                                             szs_device_scope_t device,
                                                             sz_sequence_u64tape_t const *queries,
                                                             sz_sequence_u64tape_t const *candidates,
                                                             sz_ssize_t *results, sz_size_t results_row_stride,
                                                             char const **error_message);
SZ_API_RUNTIME void szs_smith_waterman_scores_free(szs_smith_waterman_scores_t engine);

/* ---- fingerprints (stringzillas.h:532-596) - the "next" row of SURVEY.md section 8f-3 --------------------------------- */
/*  Rolling MinHash + Count-Min sketches: per dimension, the minimum over all windows of a text of a polynomial rolling
 *  hash, its low 32 bits and the number of windows that attain it - bit-exact with the reference's serial engines
 *  (include/stringzillas/fingerprints/serial.hpp:1119,646), per-dimension parameters and window widths assigned exactly as
 *  its C shim assigns them (c/stringzillas/fingerprints.cuh:31-176).
 *    dimensions        > 0; ideally a multiple of 64 x the number of window widths (one width per wavefront)
 *    alphabet_size     accepted, unused by the reference's f64 hasher (0 = 256)
 *    window_widths     NULL / 0 = {3, 4, 5, 7, 9, 11, 15, 31}; every width within [2, 65536], else sz_unexpected_dimensions_k
 *    min_hashes        `count` rows of `dimensions` u32, rows `min_hashes_stride` BYTES apart (>= 4 * dimensions, multiple of 4);
 *    min_counts        likewise.  A text shorter than a dimension's window yields hash 0xFFFFFFFF and count 0.
 *  Outputs may live in device, unified or plain host memory (the latter two are staged); text bytes must be device-accessible. */

SZ_API_RUNTIME sz_status_t szs_fingerprints_init(sz_size_t dimensions, sz_size_t alphabet_size,
                                                 sz_size_t const *window_widths, sz_size_t window_widths_count,
                                                 sz_u64_t seed, sz_memory_allocator_t const *alloc,
                                                 sz_capability_t capabilities, szs_fingerprints_t *engine,
                                                 char const **error_message);
SZ_API_RUNTIME sz_status_t szs_fingerprints_sequence(szs_fingerprints_t engine, szs_device_scope_t device,
                                                     sz_sequence_t const *texts, sz_u32_t *min_hashes,
                                                     sz_size_t min_hashes_stride, sz_u32_t *min_counts,
                                                     sz_size_t min_counts_stride, char const **error_message);
SZ_API_RUNTIME sz_status_t szs_fingerprints_u64tape(szs_fingerprints_t engine, szs_device_scope_t device,
                                                    sz_sequence_u64tape_t const *texts, sz_u32_t *min_hashes,
                                                    sz_size_t min_hashes_stride, sz_u32_t *min_counts,
                                                    sz_size_t min_counts_stride, char const **error_message);
SZ_API_RUNTIME sz_status_t szs_fingerprints_u32tape(szs_fingerprints_t engine, szs_device_scope_t device,
                                                    sz_sequence_u32tape_t const *texts, sz_u32_t *min_hashes,
                                                    sz_size_t min_hashes_stride, sz_u32_t *min_counts,
                                                    sz_size_t min_counts_stride, char const **error_message);
SZ_API_RUNTIME void szs_fingerprints_free(szs_fingerprints_t engine);

#ifdef __cplusplus
} /* extern "C" */
#endif
#endif /* STRINGZILLAS_H_ */
