/*
 *  dispatch_internal.h - what the translation units of ONE ENGINE CALL share (round 6: dispatch.c had grown to 2,150 lines).
 *
 *      dispatch.c     the call itself: inputs, buffers, the decision (tier, orientation, cell width), the launch sequence, the wait
 *                     and the profile; host-planned calls; which way a call goes
 *      ways_bytes.c   device-planned byte calls: the five ways to a call's plan (same tapes, planned inside the launch,
 *                     speculated, waited for; tiny tokens)
 *      ways_runes.c   device-planned codepoint calls: transcoded, renumbered, planned on rune counts
 *      ways_tiny.c    the tiny-token launch of either family (hip/myers_tiny.hip, hip/utf8.hip: utf8_narrow_kernel)
 *
 *  Nothing here is exported (csrc/exports.map).
 */
#ifndef SZS_DISPATCH_INTERNAL_H_
#define SZS_DISPATCH_INTERNAL_H_

#include "szs_internal.h"

typedef struct szs_call_t {
    szs_engine_s *engine;
    int device;
    hipStream_t stream;
    szs_input_t const *queries, *candidates;
    int symmetric;
    uint32_t q_count, c_count;
    void *results;
    size_t results_row_stride;
    int direct; /* kernels write the caller's matrix in place */
    void *device_results;
    size_t device_stride;
    double started, phase_started, phases[6];
    int trace;
    int ranges; /* roctx ranges currently open for this call: 0, 1 (the call) or 2 (the call and a phase) */
    char const **error_message;
} szs_call_t;

/* internal statuses: how a way says "not me" */
#define SZS_NOT_DEVICE_PLANNABLE ((sz_status_t)1) /* internal: take the host-planned path instead */
#define SZS_PLAN_VERDICTS(ENGINE) ((uint32_t *)(ENGINE)->device_fused.pointer + 48)
#define SZS_TINY_NOT_TAKEN ((sz_status_t)3) /* internal: score the call the ordinary way */
#define SZS_WAY_NOT_TAKEN ((sz_status_t)4)
#define SZS_RUNES_ARE_BYTES ((sz_status_t)2) /* internal: the corpus is ASCII - the byte engines compute the same distances */

/* dispatch.c */
sz_status_t szs_call_decide(szs_engine_s const *engine, int symmetric, int runes, int force_lanes, szs_side_stats_t const *q_stats, szs_side_stats_t const *c_stats, uint32_t const *q_variants, uint32_t const *c_variants, uint32_t const (*ranks)[SZS_PLAN_RANK_SAMPLES + 1] /* the caller's sides, or NULL: not known yet */, uint64_t cells, szs_decision_t *d, char const **error_message);
void szs_call_decide_queue(szs_engine_s const *engine, szs_decision_t *d, uint32_t const (*ranks)[SZS_PLAN_RANK_SAMPLES + 1]);
int szs_decision_has_variant_zero(szs_decision_t const *d);
int szs_decision_is_one_launch(szs_decision_t const *d);
sz_status_t szs_call_prepare(szs_engine_s *engine, szs_decision_t const *d, int device, hipStream_t stream, char const **error_message);
hipError_t szs_call_enqueue(szs_engine_s *engine, szs_decision_t const *d, int device, szs_string_ref_t const *query_refs, szs_string_ref_t const *candidate_refs, void *device_results, size_t device_stride, hipStream_t stream, szs_ref_guard_t const *guard /* refs of an earlier call: validate in the kernels; else NULL */, uint32_t *launches, uint32_t *cell_bits, sz_status_t *status, char const **error_message);
void szs_call_phase(szs_call_t *call, int index);
sz_status_t szs_call_finish(szs_call_t *call, szs_decision_t const *d, hipError_t error, sz_status_t status, uint32_t launches, uint32_t cell_bits, uint64_t query_symbols, uint64_t candidate_symbols, int *stalled);
sz_status_t szs_call_place_results(szs_call_t *call);
void szs_call_stamp_refs(szs_decision_t *remembered, void const *const data[2], void const *const offsets[2], int const wide[2], szs_plan_summary_t const *summary);
sz_status_t szs_call_reserve_device_words(szs_engine_s *engine, int device, hipStream_t stream, char const **error_message);

/* ways_tiny.c */
int szs_tiny_shaped(szs_engine_s const *engine, int symmetric, szs_side_stats_t const *queries, szs_side_stats_t const *candidates);
int szs_tiny_recently_refused(szs_engine_s *engine, uint32_t q_count, uint32_t c_count, int count_down);
sz_status_t szs_cross_tiny(szs_call_t *call, uint32_t planner_mode, szs_plan_summary_t const *seen /* or NULL */, int runes);
/* ways_bytes.c, ways_runes.c */
sz_status_t szs_cross_device_planned(szs_call_t *call);
sz_status_t szs_cross_device_planned_runes(szs_call_t *call);

#endif /* SZS_DISPATCH_INTERNAL_H_ */
