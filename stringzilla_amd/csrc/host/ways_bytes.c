/*
 *  ways_bytes.c - device-planned byte calls: five ways to a call's plan (split from dispatch.c in round 6; see dispatch_internal.h).
 */
#include "dispatch_internal.h"

#include <stdio.h>
#include <stdlib.h>
#include <string.h>

/* ---- device-planned byte calls: five ways to a call's plan, tried cheapest first ------------------------------------------------
 *
 *  (round 6: one function per way - rounds 2 to 5 grew them inside one function of 380 lines.)  Each returns the call's status,
 *  or SZS_WAY_NOT_TAKEN: this call is not one for this way (or turned out not to be: nothing real was scored) - try the next.
 */

typedef struct planned_call_t {
    szs_call_t *call;
    szs_engine_s *engine;
    szs_decision_t *remembered;  /* the engine's previous device-planned call */
    szs_plan_side_t q_side, c_side; /* the caller's sides (the same one twice for a symmetric call) */
    szs_plan_summary_t volatile *summary; /* pinned: where the device planner reports */
    int use_myers, knobs_automatic, uniform_bytes;
    unsigned myers_words;
    void const *key_data[2], *key_offsets[2]; /* what "the same tapes" means */
    int key_wide[2];
    szs_plan_summary_t seen; /* the summary of THIS call's tapes, once a planner has reported */
    int have_summary;        /* ... by a speculated plan whose launches did not hold: the refs on the device are blank */
} planned_call_t;

/** The profile of a call that was scored before its statistics were known (speculated, or planned inside its launch), and what the
 *  next call may count on: the refs on the device describe these tapes; a batch of tiny tokens goes to their kernel next time. */
static void complete_from_summary(planned_call_t *way, szs_plan_summary_t const *seen) {
    szs_engine_s *engine = way->engine;
    int const symmetric = way->call->symmetric;
    uint32_t const q_count = way->call->q_count, c_count = way->call->c_count;
    szs_rocm_call_profile_t *profile = &engine->last_profile;
    profile->cells = symmetric ? seen->symmetric_cells : seen->side[0].symbols * seen->side[1].symbols;
    profile->algorithmic_bytes = (symmetric ? ((uint64_t)q_count + 1) * seen->side[0].symbols
                                            : (uint64_t)c_count * seen->side[0].symbols + (uint64_t)q_count * seen->side[1].symbols) + profile->pairs * 16;
    profile->unique_bytes += seen->side[0].symbols + (symmetric ? 0 : seen->side[1].symbols);
    profile->longest_query = seen->side[0].longest, profile->longest_candidate = seen->side[1].longest;
    szs_call_stamp_refs(way->remembered, way->key_data, way->key_offsets, way->key_wide, seen);
    if (szs_tiny_shaped(engine, symmetric, &seen->side[0], &seen->side[1]) && !szs_tiny_recently_refused(engine, q_count, c_count, 0))
        engine->tiny_valid = 1, engine->tiny_q_count = q_count, engine->tiny_c_count = c_count;
}

/** The kernel's refs for a decision's orientation: its queries longest first, its candidates shortest first. */
static void refs_of(planned_call_t const *way, szs_decision_t const *d, szs_string_ref_t const **query_refs, szs_string_ref_t const **candidate_refs) {
    *query_refs = d->transposed ? way->c_side.descending : way->q_side.descending;
    *candidate_refs = d->transposed ? way->q_side.ascending : way->c_side.ascending;
}

/**
 *  Way 3 - the same tapes again: the refs planned for them are still on the device, no planner at all.  Every workgroup and lane
 *  of the byte kernels checks its ref against the offsets as they are NOW before it touches a string (hip/kernels.h:
 *  szs_ref_guard_t), so a tape that was rewritten in place, freed or reallocated costs one re-plan, never a wrong score or a stray
 *  read.  Only launches whose kernels carry the guard take this way: unit-cost byte queries of up to 256 bytes - ONE launch of
 *  ~0.2 ms, where 25 us of planning matter (with longer queries the guarded launches were slower than planning: 128 x 128 x 1 KB
 *  over eight lanes per pair 0.67 ms behind the guard, 0.50 ms planned - profiles/r03).
 */
static sz_status_t planned_on_the_same_tapes(planned_call_t *way) {
    szs_call_t *call = way->call;
    szs_engine_s *engine = way->engine;
    szs_decision_t *const remembered = way->remembered;
    int const symmetric = call->symmetric;
    if (!(remembered->valid && remembered->refs_current && way->knobs_automatic && szs_tuning_get(szs_knob_reuse_k) != 0 &&
          remembered->tier == SZS_TIER_LANES && remembered->use_myers && !remembered->runes && !remembered->wide_cells &&
          !szs_decision_has_variant_zero(remembered) && remembered->plan.groups_count == 1 &&
          remembered->plan.groups[0].variant == SZS_MYERS_SHORT_WORDS && remembered->q_count == call->q_count && remembered->c_count == call->c_count &&
          remembered->symmetric == symmetric && remembered->key_data[0] == way->key_data[0] && remembered->key_data[1] == way->key_data[1] &&
          remembered->key_offsets[0] == way->key_offsets[0] && remembered->key_offsets[1] == way->key_offsets[1] &&
          remembered->key_wide[0] == way->key_wide[0] && remembered->key_wide[1] == way->key_wide[1]))
        return SZS_WAY_NOT_TAKEN;
    szs_decision_t const *d = remembered;
    uint32_t volatile *const stale = (uint32_t volatile *)((char *)engine->pinned_summary.pointer + 768);
    szs_ref_guard_t guard;
    memset(&guard, 0, sizeof(guard));
    guard.enabled = 1, guard.sequence = ++engine->plan_sequence, guard.stale = (uint32_t *)stale;
    for (int role = 0; role < 2; ++role) { /* kernel roles: 0 = its queries, 1 = its candidates */
        szs_plan_side_t const *side = (role == 0) == (d->transposed == 0) ? &way->q_side : &way->c_side;
        if (symmetric) side = &way->q_side;
        guard.side[role].offsets = side->offsets, guard.side[role].base = side->base;
        guard.side[role].wide = side->wide, guard.side[role].count = side->count;
    }
    *stale = 0;
    sz_status_t status = szs_call_prepare(engine, d, call->device, call->stream, call->error_message);
    if (status != sz_success_k) return status;
    szs_call_phase(call, 2);
    uint32_t launches = 0, cell_bits = 0;
    sz_status_t enqueue_status = sz_success_k;
    hipError_t error = hipEventRecord(engine->event_start, call->stream);
    szs_string_ref_t const *query_refs, *candidate_refs;
    refs_of(way, d, &query_refs, &candidate_refs);
    if (error == hipSuccess)
        error = szs_call_enqueue(engine, d, call->device, query_refs, candidate_refs, call->device_results, call->device_stride, call->stream, &guard, &launches,
                        &cell_bits, &enqueue_status, call->error_message);
    int stalled = 0;
    engine->last_profile.planner = 3;
    status = szs_call_finish(call, d, error, enqueue_status, launches, cell_bits, d->summary.side[0].symbols, d->summary.side[1].symbols, &stalled);
    if (status != sz_success_k) return status;
    if (*stale != guard.sequence) return sz_success_k; /* every ref still described its string: scored */
    remembered->refs_current = 0;                      /* the tapes changed under the same pointers: plan them afresh */
    return SZS_WAY_NOT_TAKEN;
}

/**
 *  Way 4 - the planner INSIDE the scoring launch (round 5; hip/kernels.h: szs_fused_plan_t).  The previous call of this engine was
 *  ONE launch of the short unit-cost byte kernel and this one has the same counts: the launch goes out alone - its first two
 *  workgroups sort the two sides (what hip/planner.hip does in a launch of its own) while the others wait for the refs.  No planner
 *  launch, no kernel boundary: config 2's fresh-batch call 202 -> ~190 us.  Round 6: symmetric calls too (one side, sorted once,
 *  serves both roles) and sides of up to 16,384 strings (counted and placed in two walks over their offsets).  A batch that does not
 *  fit after all (a query beyond 256 bytes, malformed offsets) is scored as empty strings; a launch whose waiting workgroups ran out
 *  of polls scored only part of the matrix: either way the call goes on to the next way.
 */
static sz_status_t planned_inside_the_launch(planned_call_t *way) {
    szs_call_t *call = way->call;
    szs_engine_s *engine = way->engine;
    szs_decision_t *const remembered = way->remembered;
    int const symmetric = call->symmetric;
    int const knob = szs_tuning_get(szs_knob_fused_k);
    if (!(remembered->valid && !remembered->runes && remembered->tier == SZS_TIER_LANES && remembered->use_myers && !remembered->wide_cells &&
          !remembered->use_queue && szs_decision_is_one_launch(remembered) && remembered->plan.groups[0].variant == SZS_MYERS_SHORT_WORDS &&
          remembered->q_count == call->q_count && remembered->c_count == call->c_count && remembered->symmetric == symmetric &&
          call->q_count <= SZS_FUSED_MOST_STRINGS_TWO_PASSES && call->c_count <= SZS_FUSED_MOST_STRINGS_TWO_PASSES && way->knobs_automatic &&
          !way->uniform_bytes && knob != 0 && (!engine->fused_gave_up || knob == 2)))
        return SZS_WAY_NOT_TAKEN;
    szs_decision_t const *d = remembered;
    szs_fused_side_report_t volatile *const reports = (szs_fused_side_report_t volatile *)((char *)engine->pinned_summary.pointer + 1024);
    uint32_t volatile *const gave_up = (uint32_t volatile *)((char *)engine->pinned_summary.pointer + 1024 + 2 * sizeof(szs_fused_side_report_t));
    szs_fused_plan_t fused;
    memset(&fused, 0, sizeof(fused));
    fused.side[0] = d->transposed ? way->c_side : way->q_side, fused.side[1] = d->transposed ? way->q_side : way->c_side;
    if (!++engine->plan_sequence) ++engine->plan_sequence; /* never 0: the ready words start there */
    fused.sequence = engine->plan_sequence;
    fused.ready = (uint32_t *)engine->device_fused.pointer, fused.report = (szs_fused_side_report_t *)reports;
    *gave_up = 0;
    fused.gave_up = (uint32_t *)gave_up, fused.poll_budget = SZS_FUSED_POLL_BUDGET;
    if (knob == 2) fused.withhold = 1, fused.poll_budget = 64; /* testing: nobody is ever told */
    sz_status_t status = szs_call_prepare(engine, d, call->device, call->stream, call->error_message); /* buffers of the previous call: nothing to allocate */
    if (status != sz_success_k) return status;
    szs_call_phase(call, 2);
    remembered->refs_current = 0; /* the launch is about to overwrite the refs */
    /* (Measured and not kept: the launch stamping the event pair itself - hipExtLaunchKernel with a start and a stop event, no
     * records around it.  The kernel's own time reads 177.0 us instead of 180.9, but the call takes 200.7 us instead of 193.7.) */
    hipError_t error = hipEventRecord(engine->event_start, call->stream);
    uint32_t launches = 0;
    if (error == hipSuccess) {
        error = (hipError_t)szs_hip_levenshtein_myers_fused(&fused, (uint64_t *)call->device_results, call->device_stride, d->layout, call->stream);
        launches = error == hipSuccess;
    }
    engine->last_streams = 1;
    int stalled = 0;
    szs_decision_t scored = *d;
    engine->last_profile.planner = 4;
    status = szs_call_finish(call, &scored, error, sz_success_k, launches, 0, 0, 0, &stalled);
    if (status != sz_success_k) return status;
    szs_fused_side_report_t sides[2];
    memcpy(sides, (void const *)reports, sizeof(sides));
    if (*gave_up == fused.sequence) { /* a workgroup ran out of polls: whatever the reports say, not every cell was scored - the ready
                                         words are zeroed before anything waits on them again, and this engine does not try again */
        engine->fused_gave_up = 1, engine->fused_zeroed = NULL;
        return SZS_WAY_NOT_TAKEN;
    }
    if (!(sides[0].sequence == fused.sequence && !sides[0].status && !sides[0].blank &&
          (symmetric || (sides[1].sequence == fused.sequence && !sides[1].status && !sides[1].blank))))
        return SZS_WAY_NOT_TAKEN; /* not this shape after all: nothing real was scored */
    /* scored; the profile and the remembered plan take this batch's figures (caller roles again; a symmetric call has one side) */
    szs_fused_side_report_t const *const of_queries = &sides[!symmetric && d->transposed ? 1 : 0];
    szs_fused_side_report_t const *const of_candidates = symmetric ? of_queries : &sides[d->transposed ? 0 : 1];
    if (call->trace)
        for (int s = 0; s < (symmetric ? 1 : 2); ++s)
            fprintf(stderr, "fused sorter %d (10 ns ticks since it began): offsets loaded %u, positions %u, refs written %u, published %u; began %d ticks after sorter 0\n",
                    s, sides[s].ticks[1], sides[s].ticks[2], sides[s].ticks[3], sides[s].ticks[4], (int)(sides[s].ticks[0] - sides[0].ticks[0]));
    szs_plan_summary_t seen_here = remembered->summary;
    seen_here.status = 0, seen_here.speculation_held = 1, seen_here.sequence = fused.sequence;
    seen_here.side[0] = of_queries->stats, seen_here.side[1] = of_candidates->stats;
    memcpy(seen_here.rank_lengths[0], of_queries->rank_lengths, sizeof(seen_here.rank_lengths[0]));
    memcpy(seen_here.rank_lengths[1], of_candidates->rank_lengths, sizeof(seen_here.rank_lengths[1]));
    /* the lower triangle of a symmetric call: sum over i of len_i x (sum over j <= i of len_j) = ((sum len)^2 + sum len^2) / 2 */
    seen_here.symmetric_cells = symmetric ? (seen_here.side[0].symbols * seen_here.side[0].symbols + of_queries->squares) / 2 : 0;
    remembered->longest[0] = seen_here.side[0].longest, remembered->longest[1] = seen_here.side[1].longest;
    complete_from_summary(way, &seen_here);
    return szs_report(sz_success_k, call->error_message, NULL);
}

/**
 *  Way 2 - speculate: launches shaped like the previous call go in right behind the planner, which validates the shape and blanks
 *  the refs of a side that does not have it.  (Round 3: only calls of ONE launch.  The launches of a mixed-length batch leave the
 *  host one after the other, longest pairs first, and reach the device in that order; enqueued behind the planner they are all
 *  released by the same event and the device takes them as it likes - the short launch's thousands of workgroups first, the long
 *  pairs late.  Config 5: 9.68 ms speculated, 9.60 planned-and-waited-for; an eighth of it 1.95 / 1.85; codepoints 8.4 / 7.1.)
 *  Leaves `way->seen` / `way->have_summary` when the planner reported but the shape did not hold.
 */
static sz_status_t planned_and_speculated(planned_call_t *way) {
    szs_call_t *call = way->call;
    szs_engine_s *engine = way->engine;
    szs_decision_t *const remembered = way->remembered;
    int const symmetric = call->symmetric;
    if (!(remembered->valid && !remembered->runes && remembered->tier == SZS_TIER_LANES && remembered->q_count == call->q_count &&
          remembered->c_count == call->c_count && remembered->symmetric == symmetric && way->knobs_automatic && !way->uniform_bytes &&
          szs_decision_is_one_launch(remembered)))
        return SZS_WAY_NOT_TAKEN;
    szs_decision_t const *d = remembered;
    szs_plan_expectation_t expected;
    memset(&expected, 0, sizeof(expected));
    expected.enabled = 1, expected.query_side = (uint32_t)d->transposed;
    expected.longest[0] = d->longest[0], expected.longest[1] = d->longest[1];
    memcpy(expected.variant_counts, d->variant_counts, sizeof(expected.variant_counts));
    expected.sequence = ++engine->plan_sequence;
    sz_status_t status = szs_call_prepare(engine, d, call->device, call->stream, call->error_message); /* buffers of the previous call: nothing to allocate */
    if (status != sz_success_k) return status;
    szs_call_phase(call, 2);
    uint32_t launches = 0, cell_bits = 0;
    remembered->refs_current = 0; /* the planner is about to overwrite the refs */
    hipError_t error = (hipError_t)szs_hip_plan(&way->q_side, symmetric ? NULL : &way->c_side, way->myers_words, &expected, (szs_plan_summary_t *)way->summary,
                                                SZS_PLAN_VERDICTS(engine), call->stream);
    if (error != hipSuccess) return szs_report_hip(error, call->error_message); /* nothing enqueued yet */
    error = hipEventRecord(engine->event_start, call->stream);
    szs_string_ref_t const *query_refs, *candidate_refs;
    refs_of(way, d, &query_refs, &candidate_refs);
    sz_status_t enqueue_status = sz_success_k;
    if (error == hipSuccess)
        error = szs_call_enqueue(engine, d, call->device, query_refs, candidate_refs, call->device_results, call->device_stride, call->stream, NULL, &launches,
                        &cell_bits, &enqueue_status, call->error_message);
    /* the summary is read after the wait inside szs_call_finish(); profile numbers come from it, so szs_call_finish() runs on a copy of the decision
     * whose statistics are filled in afterwards */
    int stalled = 0;
    szs_decision_t scored = *d;
    engine->last_profile.planner = 2;
    status = szs_call_finish(call, &scored, error, enqueue_status, launches, cell_bits, 0, 0, &stalled);
    if (status != sz_success_k) return status;
    memcpy(&way->seen, (void const *)way->summary, sizeof(way->seen));
    way->have_summary = way->seen.sequence == expected.sequence;
    if (!(way->have_summary && !way->seen.status && way->seen.speculation_held))
        return SZS_WAY_NOT_TAKEN; /* the shape changed (or the offsets are malformed): the refs were blanked, nothing real was scored */
    complete_from_summary(way, &way->seen); /* the batch had the remembered shape and has been scored */
    return szs_report(sz_success_k, call->error_message, NULL);
}

/** Way 1 - plan on the device, wait for the summary, decide, launch (and, for a batch of tiny tokens, their launch instead). */
static sz_status_t planned_and_waited_for(planned_call_t *way) {
    szs_call_t *call = way->call;
    szs_engine_s *engine = way->engine;
    szs_decision_t *const remembered = way->remembered;
    hipStream_t const stream = call->stream;
    int const symmetric = call->symmetric;
    char const **error_message = call->error_message;
    szs_plan_summary_t *const seen = &way->seen;
    sz_status_t status;
    hipError_t error;
    remembered->refs_current = 0;
    if (!way->have_summary) {
        szs_plan_expectation_t none;
        memset(&none, 0, sizeof(none));
        none.sequence = ++engine->plan_sequence;
        error = (hipError_t)szs_hip_plan(&way->q_side, symmetric ? NULL : &way->c_side, way->myers_words, &none, (szs_plan_summary_t *)way->summary,
                                         SZS_PLAN_VERDICTS(engine), stream);
        hipError_t const drained = hipStreamSynchronize(stream);
        if (error == hipSuccess) error = drained;
        if (error != hipSuccess) return szs_report_hip(error, error_message);
        memcpy(seen, (void const *)way->summary, sizeof(*seen));
        if (seen->sequence != none.sequence) return szs_report(sz_status_unknown_k, error_message, "The device planner did not report");
    }
    if (seen->status & SZS_PLAN_STATUS_DESCENDING) return szs_report(sz_unexpected_dimensions_k, error_message, "Tape offsets must ascend");
    if (seen->status & SZS_PLAN_STATUS_OVERFLOW) return szs_report(sz_overflow_risk_k, error_message, NULL);
    if (seen->status & SZS_PLAN_STATUS_UNSORTED) return SZS_NOT_DEVICE_PLANNABLE; /* strings beyond the planner's histogram */
    if (way->uniform_bytes) { /* the scan has landed (the planner's wait covered it): number the bytes that occur 0 ... A - 1 */
        uint32_t volatile const *const presence = (uint32_t volatile const *)((char *)engine->pinned_summary.pointer + 896);
        uint32_t classes = 0;
        for (unsigned byte = 0; byte < 256; ++byte)
            engine->uniform_byte_to_class[byte] = (presence[byte / 32] >> (byte % 32)) & 1u ? (uint8_t)classes++ : 0;
        engine->uniform_classes = classes ? classes : 1; /* a batch of empty strings: one class nobody belongs to */
    }
    szs_call_phase(call, 1);

    if (way->use_myers && szs_tuning_get(szs_knob_tier_k) < 0 && szs_tuning_get(szs_knob_swap_k) < 0 && szs_tuning_get(szs_knob_queue_k) < 0 &&
        szs_tiny_shaped(engine, symmetric, &seen->side[0], &seen->side[1]) && !szs_tiny_recently_refused(engine, call->q_count, call->c_count, 1)) {
        /* the summary says tiny tokens (and the kernel did not refuse the previous batch of these counts): no refs needed after all */
        status = szs_cross_tiny(call, 1, seen, 0);
        if (status != SZS_TINY_NOT_TAKEN) return status;
    }

    for (int attempt = 0; attempt < 2; ++attempt) { /* second round: a stalled band chain is re-run on the lanes tier */
        szs_decision_t d;
        uint64_t const cells = symmetric ? seen->symmetric_cells : seen->side[0].symbols * seen->side[1].symbols;
        status = szs_call_decide(engine, symmetric, 0, attempt > 0, &seen->side[0], &seen->side[1], seen->variant_counts[0], seen->variant_counts[1],
                        seen->rank_lengths, cells, &d, error_message);
        if (status != sz_success_k) return status;
        szs_call_decide_queue(engine, &d, seen->rank_lengths);
        status = szs_call_prepare(engine, &d, call->device, stream, error_message);
        if (status != sz_success_k) return status;
        szs_call_phase(call, 2);
        if (way->have_summary) { /* the refs on the device are blank (failed speculation): write the real ones */
            szs_plan_expectation_t none;
            memset(&none, 0, sizeof(none));
            none.sequence = ++engine->plan_sequence;
            error = (hipError_t)szs_hip_plan(&way->q_side, symmetric ? NULL : &way->c_side, way->myers_words, &none, (szs_plan_summary_t *)way->summary,
                                             SZS_PLAN_VERDICTS(engine), stream);
            if (error != hipSuccess) return szs_report_hip(error, error_message);
            way->have_summary = 0;
        }
        uint32_t launches = 0, cell_bits = 0;
        error = hipEventRecord(engine->event_start, stream);
        szs_string_ref_t const *query_refs, *candidate_refs;
        refs_of(way, &d, &query_refs, &candidate_refs);
        sz_status_t enqueue_status = sz_success_k;
        if (error == hipSuccess)
            error = szs_call_enqueue(engine, &d, call->device, query_refs, candidate_refs, call->device_results, call->device_stride, stream, NULL, &launches,
                            &cell_bits, &enqueue_status, error_message);
        int stalled = 0;
        engine->last_profile.planner = 1;
        status = szs_call_finish(call, &d, error, enqueue_status, launches, cell_bits, seen->side[0].symbols, seen->side[1].symbols, &stalled);
        if (status != sz_success_k) return status;
        if (!stalled) {
            *remembered = d; /* the next call of this shape goes in speculatively - or, on the same tapes, without a planner */
            szs_call_stamp_refs(remembered, way->key_data, way->key_offsets, way->key_wide, seen);
            return sz_success_k;
        }
    }
    return szs_report(sz_status_unknown_k, error_message, "Systolic pipeline stalled");
}

sz_status_t szs_cross_device_planned(szs_call_t *call) {
    szs_engine_s *engine = call->engine;
    hipStream_t const stream = call->stream;
    int const device = call->device, symmetric = call->symmetric;
    uint32_t const q_count = call->q_count, c_count = call->c_count;
    char const **error_message = call->error_message;

    size_t const refs_bytes = 2 * ((size_t)q_count + (symmetric ? 0 : c_count)) * sizeof(szs_string_ref_t);
    void *const refs_before = engine->device_plan_refs.pointer;
    sz_status_t status = szs_buffer_reserve(&engine->device_plan_refs, szs_memory_device_k, device, refs_bytes, error_message);
    /* The refs of the previous call live in that buffer.  If the reserve moved it (or failed), the remembered plan describes
     * memory that is gone: forget it HERE, before any way below could re-use it behind nothing but the in-kernel guard. */
    if (engine->remembered && (status != sz_success_k || engine->device_plan_refs.pointer != refs_before))
        engine->remembered->refs_current = 0, engine->remembered->valid = 0;
    if (status != sz_success_k) return status;
    if (!engine->remembered) {
        engine->remembered = (szs_decision_t *)calloc(1, sizeof(szs_decision_t));
        if (!engine->remembered) return szs_report(sz_bad_alloc_k, error_message, NULL);
    }
    planned_call_t way;
    memset(&way, 0, sizeof(way));
    way.call = call, way.engine = engine, way.remembered = engine->remembered;
    szs_string_ref_t *const base = (szs_string_ref_t *)engine->device_plan_refs.pointer;
    szs_plan_side_t const q_side = {call->queries->offsets, (uint64_t)(uintptr_t)call->queries->data, q_count,
                                    call->queries->kind == szs_input_u64tape_k, base, base + q_count, NULL, NULL};
    way.q_side = way.c_side = q_side;
    if (!symmetric) {
        szs_plan_side_t const other = {call->candidates->offsets, (uint64_t)(uintptr_t)call->candidates->data, c_count,
                                       call->candidates->kind == szs_input_u64tape_k, base + 2 * (size_t)q_count,
                                       base + 2 * (size_t)q_count + c_count, NULL, NULL};
        way.c_side = other;
    }
    way.summary = (szs_plan_summary_t volatile *)engine->pinned_summary.pointer;
    /* (the codepoint family gets here with an ASCII corpus: its runes are its bytes) */
    way.use_myers = engine->is_unit_cost && (engine->family == szs_family_levenshtein_k || engine->family == szs_family_levenshtein_utf8_k);
    way.myers_words = way.use_myers ? SZS_MYERS_MAX_WORDS : 0;
    status = szs_call_place_results(call);
    if (status != sz_success_k) return status;
    status = szs_call_reserve_device_words(engine, device, stream, error_message);
    if (status != sz_success_k) return status;
    szs_call_phase(call, 0);

    /* ---- way 5, tiny tokens (hip/myers_tiny.hip): the previous call of these counts was scored straight from the tapes - so is this
     * one, with no planner at all; the kernel says when a query does not fit it */
    if (engine->tiny_valid && engine->tiny_q_count == q_count && engine->tiny_c_count == c_count && way.use_myers &&
        szs_tuning_get(szs_knob_tiny_k) != 0 && szs_tuning_get(szs_knob_speculate_k) != 0 && szs_tuning_get(szs_knob_tier_k) < 0 &&
        szs_tuning_get(szs_knob_swap_k) < 0 && szs_tuning_get(szs_knob_queue_k) < 0) {
        status = szs_cross_tiny(call, 5, NULL, 0);
        if (status != SZS_TINY_NOT_TAKEN) return status;
    }

    way.knobs_automatic = szs_tuning_get(szs_knob_speculate_k) != 0 && szs_tuning_get(szs_knob_tier_k) < 0 && szs_tuning_get(szs_knob_swap_k) < 0 &&
                          szs_tuning_get(szs_knob_cells_k) < 0 && szs_tuning_get(szs_knob_packed_k) < 0 && szs_tuning_get(szs_knob_team_k) < 0 &&
                          szs_tuning_get(szs_knob_queue_k) < 0;
    way.key_data[0] = call->queries->data, way.key_data[1] = symmetric ? call->queries->data : call->candidates->data;
    way.key_offsets[0] = call->queries->offsets, way.key_offsets[1] = symmetric ? call->queries->offsets : call->candidates->offsets;
    way.key_wide[0] = (int)way.q_side.wide, way.key_wide[1] = (int)way.c_side.wide;

    status = planned_on_the_same_tapes(&way); /* way 3 */
    if (status != SZS_WAY_NOT_TAKEN) return status;

    /* ---- a Levenshtein engine with non-unit costs: which bytes occur in this batch?  One pass over both tapes, enqueued ahead
     * of the planner and read after the planner's own wait; the team tier keys its profile by the classes the host numbers
     * from it (szs_call_decide()).  Such a call is not speculated: its launch depends on what the scan finds. */
    way.uniform_bytes = (engine->family == szs_family_levenshtein_k || engine->family == szs_family_levenshtein_utf8_k) && !engine->is_unit_cost &&
                        szs_tuning_get(szs_knob_packed_k) != 0 && szs_tuning_get(szs_knob_team_k) != 0;
    engine->uniform_classes = 0;
    if (way.uniform_bytes) {
        uint32_t volatile *const presence = (uint32_t volatile *)((char *)engine->pinned_summary.pointer + 896);
        status = szs_buffer_reserve(&engine->device_presence, szs_memory_device_k, device, 8 * sizeof(uint32_t), error_message);
        if (status != sz_success_k) return status;
        hipError_t error = hipMemsetAsync(engine->device_presence.pointer, 0, 8 * sizeof(uint32_t), stream);
        if (error == hipSuccess)
            error = (hipError_t)szs_hip_byte_presence(call->queries->data, call->queries->offsets, q_count, (int)way.q_side.wide,
                                                      (uint32_t *)engine->device_presence.pointer, stream);
        if (error == hipSuccess && !symmetric)
            error = (hipError_t)szs_hip_byte_presence(call->candidates->data, call->candidates->offsets, c_count, (int)way.c_side.wide,
                                                      (uint32_t *)engine->device_presence.pointer, stream);
        if (error == hipSuccess)
            error = hipMemcpyAsync((void *)presence, engine->device_presence.pointer, 8 * sizeof(uint32_t), hipMemcpyDeviceToHost, stream);
        if (error != hipSuccess) {
            (void)hipStreamSynchronize(stream);
            return szs_report_hip(error, error_message);
        }
    }

    status = planned_inside_the_launch(&way); /* way 4 */
    if (status != SZS_WAY_NOT_TAKEN) return status;
    status = planned_and_speculated(&way); /* way 2 */
    if (status != SZS_WAY_NOT_TAKEN) return status;
    return planned_and_waited_for(&way); /* way 1 */
}

