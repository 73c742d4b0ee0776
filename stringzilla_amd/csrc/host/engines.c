/*
 *  engines.c - the `extern "C"` engine entry points: init / call (sequence, u32tape, u64tape) / free for the four
 *  similarity families and for the fingerprint engines, and the ROCm profile accessor.
 *
 *  ROCm counterpart of c/stringzillas/{levenshtein,needleman_wunsch,smith_waterman,fingerprints}.cuh.
 *  The capability ladder of the reference (levenshtein.cuh:107-200) collapses to one rung here: this build ships
 *  GPU engines only, so `capabilities` must contain sz_cap_cuda_k.
 */
#include "szs_internal.h"

#include <stdlib.h>
#include <string.h>

void szs_engine_release(szs_engine_s *engine); /* dispatch.c */

static unsigned magnitude_of(int value) { return (unsigned)(value < 0 ? -value : value); }

static sz_status_t engine_new(szs_family_t family, sz_capability_t capabilities, void **out, szs_engine_s **created,
                              char const **error_message) {
    if (!out) return szs_report(sz_status_unknown_k, error_message, "Engine must not be null");
    if (*out) return szs_report(sz_status_unknown_k, error_message, "Engine must be uninitialized");
    /* Strict by default; with the `cpu_requests` knob set to "gpu" a mask without the GPU bit is served by the only engines
     * this build has (host/tuning.c) - the results are the same numbers, computed on the GPU. */
    if ((capabilities & sz_cap_cuda_k) == 0 && szs_tuning_get(szs_knob_cpu_requests_k) != 1)
        return szs_report(sz_missing_gpu_k, error_message,
                          "The ROCm build ships GPU engines only: request sz_cap_cuda_k (e.g. from a GPU device scope)");
    if ((szs_capabilities() & sz_cap_cuda_k) == 0) return szs_report(sz_missing_gpu_k, error_message, NULL);
    szs_engine_s *engine = (szs_engine_s *)calloc(1, sizeof(szs_engine_s));
    if (!engine) return szs_report(sz_bad_alloc_k, error_message, NULL);
    engine->magic = SZS_ENGINE_MAGIC;
    engine->family = family;
    engine->device = -1, engine->events_device = -1, engine->model_uploaded_device = -1, engine->aux_device = -1;
    *created = engine;
    *out = engine;
    return szs_report(sz_success_k, error_message, NULL);
}

static void engine_free(void *handle) {
    szs_engine_s *engine = (szs_engine_s *)handle;
    if (!engine || engine->magic != SZS_ENGINE_MAGIC) return;
    szs_engine_release(engine);
    engine->magic = 0;
    free(engine);
}

static sz_status_t levenshtein_init(szs_family_t family, sz_error_cost_t match, sz_error_cost_t mismatch,
                                    sz_error_cost_t open, sz_error_cost_t extend, sz_capability_t capabilities,
                                    void **out, char const **error_message) {
    szs_engine_s *engine = NULL;
    sz_status_t const status = engine_new(family, capabilities, out, &engine, error_message);
    if (status != sz_success_k) return status;
    engine->match = match, engine->mismatch = mismatch, engine->open = open, engine->extend = extend;
    engine->is_linear = open == extend;                                               /* levenshtein.cuh:117 */
    engine->is_unit_cost = match == 0 && mismatch == 1 && open == 1 && extend == 1;   /* serial.hpp:118-120 */
    unsigned magnitude = magnitude_of(match);
    if (magnitude_of(mismatch) > magnitude) magnitude = magnitude_of(mismatch);
    if (magnitude_of(open) > magnitude) magnitude = magnitude_of(open);
    if (magnitude_of(extend) > magnitude) magnitude = magnitude_of(extend);
    engine->magnitude = magnitude;
    return sz_success_k;
}

static sz_status_t scores_init(szs_family_t family, sz_u8_t const *byte_to_class, sz_error_cost_t const *class_costs,
                               sz_error_cost_t open, sz_error_cost_t extend, sz_capability_t capabilities, void **out,
                               char const **error_message) {
    if (!byte_to_class || !class_costs)
        return szs_report(sz_status_unknown_k, error_message, "Substitution tables must not be null");
    szs_engine_s *engine = NULL;
    sz_status_t const status = engine_new(family, capabilities, out, &engine, error_message);
    if (status != sz_success_k) return status;
    memcpy(engine->byte_to_class, byte_to_class, 256); /* needleman_wunsch.cuh:99-118: both tables are copied */
    memcpy(engine->class_costs, class_costs, 32 * 32);
    for (int i = 0; i < 256; ++i) engine->byte_to_class[i] &= 31; /* a class is one of 32 */
    engine->open = open, engine->extend = extend;
    engine->is_linear = open == extend;
    unsigned magnitude = magnitude_of(open) > magnitude_of(extend) ? magnitude_of(open) : magnitude_of(extend);
    for (int i = 0; i < 32 * 32; ++i)
        if (magnitude_of(class_costs[i]) > magnitude) magnitude = magnitude_of(class_costs[i]);
    engine->magnitude = magnitude;
    return sz_success_k;
}

static szs_input_t input_from_sequence(sz_sequence_t const *sequence) {
    szs_input_t input = {szs_input_sequence_k, sequence->count, NULL, NULL, sequence};
    return input;
}
static szs_input_t input_from_u32tape(sz_sequence_u32tape_t const *tape) {
    szs_input_t input = {szs_input_u32tape_k, tape->count, tape->data, tape->offsets, NULL};
    return input;
}
static szs_input_t input_from_u64tape(sz_sequence_u64tape_t const *tape) {
    szs_input_t input = {szs_input_u64tape_k, tape->count, tape->data, tape->offsets, NULL};
    return input;
}

#define SZS_CROSS_BODY(MAKE_INPUT)                                                                                     \
    if (!queries) return szs_report(sz_status_unknown_k, error_message, "Queries must not be null");                   \
    szs_input_t const query_input = MAKE_INPUT(queries);                                                               \
    szs_input_t candidate_input;                                                                                       \
    if (candidates) candidate_input = MAKE_INPUT(candidates);                                                          \
    return szs_engine_cross((szs_engine_s *)engine, (szs_scope_s *)device, &query_input,                               \
                            candidates ? &candidate_input : NULL, results, results_row_stride, error_message);

/* ---- Levenshtein, bytes (stringzillas.h:197-254) -------------------------------------------------------------------- */

sz_status_t szs_levenshtein_distances_init(sz_error_cost_t match, sz_error_cost_t mismatch, sz_error_cost_t open,
                                           sz_error_cost_t extend, sz_memory_allocator_t const *alloc,
                                           sz_capability_t capabilities, szs_levenshtein_distances_t *engine,
                                           char const **error_message) {
    (void)alloc; /* accepted and ignored, as in the reference (levenshtein.cuh:112) */
    return levenshtein_init(szs_family_levenshtein_k, match, mismatch, open, extend, capabilities, engine, error_message);
}
sz_status_t szs_levenshtein_distances(szs_levenshtein_distances_t engine, szs_device_scope_t device,
                                      sz_sequence_t const *queries, sz_sequence_t const *candidates, sz_size_t *results,
                                      sz_size_t results_row_stride, char const **error_message) {
    SZS_CROSS_BODY(input_from_sequence)
}
sz_status_t szs_levenshtein_distances_u32tape(szs_levenshtein_distances_t engine, szs_device_scope_t device,
                                              sz_sequence_u32tape_t const *queries,
                                              sz_sequence_u32tape_t const *candidates, sz_size_t *results,
                                              sz_size_t results_row_stride, char const **error_message) {
    SZS_CROSS_BODY(input_from_u32tape)
}
sz_status_t szs_levenshtein_distances_u64tape(szs_levenshtein_distances_t engine, szs_device_scope_t device,
                                              sz_sequence_u64tape_t const *queries,
                                              sz_sequence_u64tape_t const *candidates, sz_size_t *results,
                                              sz_size_t results_row_stride, char const **error_message) {
    SZS_CROSS_BODY(input_from_u64tape)
}
void szs_levenshtein_distances_free(szs_levenshtein_distances_t engine) { engine_free(engine); }

/* ---- Levenshtein, UTF-8 codepoints (stringzillas.h:271-328) - a "next" row of SURVEY.md section 8f ----------------- */

sz_status_t szs_levenshtein_distances_utf8_init(sz_error_cost_t match, sz_error_cost_t mismatch, sz_error_cost_t open,
                                                sz_error_cost_t extend, sz_memory_allocator_t const *alloc,
                                                sz_capability_t capabilities, szs_levenshtein_distances_utf8_t *engine,
                                                char const **error_message) {
    (void)alloc;
    return levenshtein_init(szs_family_levenshtein_utf8_k, match, mismatch, open, extend, capabilities, engine,
                            error_message);
}
sz_status_t szs_levenshtein_distances_utf8(szs_levenshtein_distances_utf8_t engine, szs_device_scope_t device,
                                           sz_sequence_t const *queries, sz_sequence_t const *candidates,
                                           sz_size_t *results, sz_size_t results_row_stride,
                                           char const **error_message) {
    SZS_CROSS_BODY(input_from_sequence)
}
sz_status_t szs_levenshtein_distances_utf8_u32tape(szs_levenshtein_distances_utf8_t engine, szs_device_scope_t device,
                                                   sz_sequence_u32tape_t const *queries,
                                                   sz_sequence_u32tape_t const *candidates, sz_size_t *results,
                                                   sz_size_t results_row_stride, char const **error_message) {
    SZS_CROSS_BODY(input_from_u32tape)
}
sz_status_t szs_levenshtein_distances_utf8_u64tape(szs_levenshtein_distances_utf8_t engine, szs_device_scope_t device,
                                                   sz_sequence_u64tape_t const *queries,
                                                   sz_sequence_u64tape_t const *candidates, sz_size_t *results,
                                                   sz_size_t results_row_stride, char const **error_message) {
    SZS_CROSS_BODY(input_from_u64tape)
}
void szs_levenshtein_distances_utf8_free(szs_levenshtein_distances_utf8_t engine) { engine_free(engine); }

/* ---- Needleman-Wunsch (stringzillas.h:355-413) ----------------------------------------------------------------------- */

sz_status_t szs_needleman_wunsch_scores_init(sz_u8_t const *byte_to_class, sz_error_cost_t const *class_substitution_costs,
                                             sz_error_cost_t open, sz_error_cost_t extend,
                                             sz_memory_allocator_t const *alloc, sz_capability_t capabilities,
                                             szs_needleman_wunsch_scores_t *engine, char const **error_message) {
    (void)alloc;
    return scores_init(szs_family_needleman_wunsch_k, byte_to_class, class_substitution_costs, open, extend,
                       capabilities, engine, error_message);
}
sz_status_t szs_needleman_wunsch_scores(szs_needleman_wunsch_scores_t engine, szs_device_scope_t device,
                                        sz_sequence_t const *queries, sz_sequence_t const *candidates,
                                        sz_ssize_t *results, sz_size_t results_row_stride, char const **error_message) {
    SZS_CROSS_BODY(input_from_sequence)
}
sz_status_t szs_needleman_wunsch_scores_u32tape(szs_needleman_wunsch_scores_t engine, szs_device_scope_t device,
                                                sz_sequence_u32tape_t const *queries,
                                                sz_sequence_u32tape_t const *candidates, sz_ssize_t *results,
                                                sz_size_t results_row_stride, char const **error_message) {
    SZS_CROSS_BODY(input_from_u32tape)
}
sz_status_t szs_needleman_wunsch_scores_u64tape(szs_needleman_wunsch_scores_t engine, szs_device_scope_t device,
                                                sz_sequence_u64tape_t const *queries,
                                                sz_sequence_u64tape_t const *candidates, sz_ssize_t *results,
                                                sz_size_t results_row_stride, char const **error_message) {
    SZS_CROSS_BODY(input_from_u64tape)
}
void szs_needleman_wunsch_scores_free(szs_needleman_wunsch_scores_t engine) { engine_free(engine); }

/* ---- Smith-Waterman (stringzillas.h:430-488) ------------------------------------------------------------------------- */

sz_status_t szs_smith_waterman_scores_init(sz_u8_t const *byte_to_class, sz_error_cost_t const *class_substitution_costs,
                                           sz_error_cost_t open, sz_error_cost_t extend,
                                           sz_memory_allocator_t const *alloc, sz_capability_t capabilities,
                                           szs_smith_waterman_scores_t *engine, char const **error_message) {
    (void)alloc;
    return scores_init(szs_family_smith_waterman_k, byte_to_class, class_substitution_costs, open, extend, capabilities,
                       engine, error_message);
}
sz_status_t szs_smith_waterman_scores(szs_smith_waterman_scores_t engine, szs_device_scope_t device,
                                      sz_sequence_t const *queries, sz_sequence_t const *candidates, sz_ssize_t *results,
                                      sz_size_t results_row_stride, char const **error_message) {
    SZS_CROSS_BODY(input_from_sequence)
}
sz_status_t szs_smith_waterman_scores_u32tape(szs_smith_waterman_scores_t engine, szs_device_scope_t device,
                                              sz_sequence_u32tape_t const *queries,
                                              sz_sequence_u32tape_t const *candidates, sz_ssize_t *results,
                                              sz_size_t results_row_stride, char const **error_message) {
    SZS_CROSS_BODY(input_from_u32tape)
}
sz_status_t szs_smith_waterman_scores_u64tape(szs_smith_waterman_scores_t engine, szs_device_scope_t device,
                                              sz_sequence_u64tape_t const *queries,
                                              sz_sequence_u64tape_t const *candidates, sz_ssize_t *results,
                                              sz_size_t results_row_stride, char const **error_message) {
    SZS_CROSS_BODY(input_from_u64tape)
}
void szs_smith_waterman_scores_free(szs_smith_waterman_scores_t engine) { engine_free(engine); }

/* ---- fingerprints (stringzillas.h:532-596): host/fingerprint_engines.c + hip/fingerprints.hip -------------------------------- */

sz_status_t szs_fingerprints_init(sz_size_t dimensions, sz_size_t alphabet_size, sz_size_t const *window_widths,
                                  sz_size_t window_widths_count, sz_u64_t seed, sz_memory_allocator_t const *alloc,
                                  sz_capability_t capabilities, szs_fingerprints_t *engine, char const **error_message) {
    (void)alloc; /* accepted and ignored, as in the reference (fingerprints.cuh:38) */
    return szs_fingerprints_create(dimensions, alphabet_size, window_widths, window_widths_count, seed, capabilities, engine,
                                   error_message);
}
sz_status_t szs_fingerprints_sequence(szs_fingerprints_t engine, szs_device_scope_t device, sz_sequence_t const *texts,
                                      sz_u32_t *min_hashes, sz_size_t min_hashes_stride, sz_u32_t *min_counts,
                                      sz_size_t min_counts_stride, char const **error_message) {
    if (!texts) return szs_report(sz_status_unknown_k, error_message, "Input texts cannot be null");
    szs_input_t const input = input_from_sequence(texts);
    return szs_fingerprints_call((szs_fingerprints_s *)engine, (szs_scope_s *)device, &input, min_hashes, min_hashes_stride,
                                 min_counts, min_counts_stride, error_message);
}
sz_status_t szs_fingerprints_u64tape(szs_fingerprints_t engine, szs_device_scope_t device,
                                     sz_sequence_u64tape_t const *texts, sz_u32_t *min_hashes,
                                     sz_size_t min_hashes_stride, sz_u32_t *min_counts, sz_size_t min_counts_stride,
                                     char const **error_message) {
    if (!texts) return szs_report(sz_status_unknown_k, error_message, "Input texts cannot be null");
    szs_input_t const input = input_from_u64tape(texts);
    return szs_fingerprints_call((szs_fingerprints_s *)engine, (szs_scope_s *)device, &input, min_hashes, min_hashes_stride,
                                 min_counts, min_counts_stride, error_message);
}
sz_status_t szs_fingerprints_u32tape(szs_fingerprints_t engine, szs_device_scope_t device,
                                     sz_sequence_u32tape_t const *texts, sz_u32_t *min_hashes,
                                     sz_size_t min_hashes_stride, sz_u32_t *min_counts, sz_size_t min_counts_stride,
                                     char const **error_message) {
    if (!texts) return szs_report(sz_status_unknown_k, error_message, "Input texts cannot be null");
    szs_input_t const input = input_from_u32tape(texts);
    return szs_fingerprints_call((szs_fingerprints_s *)engine, (szs_scope_s *)device, &input, min_hashes, min_hashes_stride,
                                 min_counts, min_counts_stride, error_message);
}
void szs_fingerprints_free(szs_fingerprints_t engine) { szs_fingerprints_destroy((szs_fingerprints_s *)engine); }

/* ---- ROCm extension --------------------------------------------------------------------------------------------------- */

sz_status_t szs_rocm_last_call_profile(void *handle, szs_rocm_call_profile_t *profile) {
    szs_engine_s *engine = (szs_engine_s *)handle;
    if (!engine || engine->magic != SZS_ENGINE_MAGIC || !profile) return sz_status_unknown_k;
    *profile = engine->last_profile;
    return sz_success_k;
}
