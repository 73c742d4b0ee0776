/*
 *  ways_runes.c - device-planned codepoint calls (split from dispatch.c in round 6; see dispatch_internal.h).
 */
#include "dispatch_internal.h"

#include <stdio.h>
#include <stdlib.h>
#include <string.h>

/* ---- device-planned codepoint calls --------------------------------------------------------------------------------------- */


/**
 *  The codepoint engine over tapes the device can read, without the host reading a single offset (round 2 planned these calls
 *  on the host: offsets downloaded, strings gathered and re-addressed in O(Q + C) host loops, a wait between transcoding and
 *  planning - a third of the wall time of a batch of short words).  One stream, one wait:
 *      transcode both tapes (hip/utf8.hip: rune starts follow from the byte offsets alone, no scan) -> renumber the runes
 *      -> plan on RUNE counts (hip/planner.hip) -> wait -> decide -> launch.
 *  The UTF-32 buffer is sized by the previous calls; a batch that needs more says so (`needed`) and is transcoded again.
 *  An ASCII corpus goes to the byte engines (serial.hpp:2809-2813, applied per call).
 */
/** Both tapes into the engine's UTF-32 buffer and, with `renumber`, their runes into ids: launches only, no wait. */
static hipError_t enqueue_transcoding(szs_call_t *call, char *remote, size_t flags_at, size_t needed_at, size_t staging_bytes, uint64_t *starts,
                                      uint32_t *counts, int renumber) {
    szs_engine_s *engine = call->engine;
    hipStream_t const stream = call->stream;
    uint32_t const q_count = call->q_count, c_count = call->c_count;
    uint64_t const capacity = engine->device_runes.capacity / sizeof(uint32_t);
    uint32_t *const device_flags = (uint32_t *)(remote + flags_at);
    size_t const strings = (size_t)q_count + (call->symmetric ? 0 : c_count);
    hipError_t error = hipMemsetAsync(remote + flags_at, 0, staging_bytes - flags_at, stream);
    if (error == hipSuccess)
        error = (hipError_t)szs_hip_utf8_transcode_tapes(call->queries->data, call->queries->offsets, q_count, call->queries->kind == szs_input_u64tape_k,
                                                         call->symmetric ? NULL : call->candidates->data,
                                                         call->symmetric ? NULL : call->candidates->offsets, call->symmetric ? 0u : c_count,
                                                         !call->symmetric && call->candidates->kind == szs_input_u64tape_k, capacity,
                                                         (uint32_t *)engine->device_runes.pointer, starts, counts, device_flags,
                                                         (uint64_t *)(remote + needed_at), renumber ? engine->device_alphabet.pointer : NULL, stream);
    if (error == hipSuccess && renumber)
        error = (hipError_t)szs_hip_alphabet_rename((uint32_t)strings, starts, counts, (uint32_t *)engine->device_runes.pointer, device_flags,
                                                    engine->device_alphabet.pointer, 1, SZS_ALPHABET_MOST, device_flags + 1, stream);
    return error;
}

/** The size of the direct tables the codepoint kernels are launched with for a batch of `distinct` renumbered runes: some
 *  room above it, so that the NEXT batch of the stream - launched on this one's shape before anyone has counted its runes -
 *  still fits when it holds a few more (a table row is 4 bytes of LDS). */
static uint32_t alphabet_with_room(uint32_t distinct) {
    uint32_t const roomy = distinct + distinct / 8 + 8;
    return roomy < SZS_ALPHABET_MOST ? roomy : SZS_ALPHABET_MOST;
}

sz_status_t szs_cross_device_planned_runes(szs_call_t *call) {
    szs_engine_s *engine = call->engine;
    hipStream_t const stream = call->stream;
    int const device = call->device, symmetric = call->symmetric;
    uint32_t const q_count = call->q_count, c_count = call->c_count;
    char const **error_message = call->error_message;
    size_t const strings = (size_t)q_count + (symmetric ? 0 : c_count);

    /* device staging: [rune starts, u64][rune counts, u32][any_multibyte, distinct runes, alphabet overflow, pad][needed, u64] */
    size_t const starts_at = 0, counts_at = strings * sizeof(uint64_t);
    size_t const flags_at = (counts_at + strings * sizeof(uint32_t) + 7) & ~(size_t)7, needed_at = flags_at + 4 * sizeof(uint32_t);
    size_t const staging_bytes = needed_at + sizeof(uint64_t);
    /* (... or, for a batch of tiny tokens - cross_tiny: a word per string) */
    size_t const narrow_staging_bytes = strings * sizeof(uint64_t);
    sz_status_t status = szs_buffer_reserve(&engine->device_transcode, szs_memory_device_k, device,
                                            staging_bytes > narrow_staging_bytes ? staging_bytes : narrow_staging_bytes, error_message);
    if (status == sz_success_k) status = szs_buffer_reserve(&engine->pinned_transcode, szs_memory_pinned_k, device, 64, error_message);
    void *const refs_before = engine->device_plan_refs.pointer;
    if (status == sz_success_k)
        status = szs_buffer_reserve(&engine->device_plan_refs, szs_memory_device_k, device, 2 * strings * sizeof(szs_string_ref_t), error_message);
    if (engine->remembered && (status != sz_success_k || engine->device_plan_refs.pointer != refs_before))
        engine->remembered->refs_current = 0, engine->remembered->valid = 0;
    if (status == sz_success_k && engine->device_runes.capacity < ((size_t)1 << 20))
        status = szs_buffer_reserve(&engine->device_runes, szs_memory_device_k, device, (size_t)1 << 20, error_message);
    /* Renumbering the runes (hip/utf8.hip) is four more operations ahead of the planner - ~60 us and a pass over every rune,
     * ~15 ps each - and makes the scoring kernels ~15 % faster (one LDS read per column instead of a hash probe, ~3 fs per
     * cell): worth it when the CELLS of the call outweigh its runes.  4096 x 4096 words of prose (6e8 cells): 0.39 ms
     * renumbered, 0.31 not; config 5u (4.4e11 cells): 7.1 against 8.4 ms, an eighth of it 1.82 / 2.16.  The host has not read
     * an offset, so it goes by the PREVIOUS call of this engine - a stream of batches settles at once. */
    int const alphabet_knob = szs_tuning_get(szs_knob_alphabet_k);
    int const renumber = alphabet_knob == 0  ? 0
                         : alphabet_knob > 0 ? 1
                                             : engine->cells_before >= 20000000000ull + 5000ull * engine->runes_needed && engine->runes_needed > 0;
    if (status == sz_success_k && renumber)
        status = szs_buffer_reserve(&engine->device_alphabet, szs_memory_device_k, device, szs_hip_alphabet_workspace_bytes(), error_message);
    if (status == sz_success_k) status = szs_call_place_results(call);
    if (status == sz_success_k) status = szs_call_reserve_device_words(engine, device, stream, error_message);
    if (status == sz_success_k && !engine->remembered) {
        engine->remembered = (szs_decision_t *)calloc(1, sizeof(szs_decision_t));
        if (!engine->remembered) status = szs_report(sz_bad_alloc_k, error_message, NULL);
    }
    if (status != sz_success_k) return status;
    szs_decision_t *const remembered = engine->remembered;
    remembered->refs_current = 0; /* the planner is about to overwrite the refs */
    szs_call_phase(call, 0);

    /* ---- tiny tokens (round 6): the previous call of these counts was words of a few runes - this one is narrowed to byte strings and
     * scored by the tiny-token launch without being transcoded, renumbered or planned (cross_tiny; the byte path's way 5).  A batch that
     * is something else says so itself (a string beyond 255 runes, too many long ones, an alphabet beyond the table) and is scored below. */
    /* (... or was ASCII words, which this engine hands to the byte kernels - after transcoding and planning them to find that out:
     * narrowed, an ASCII batch is its own bytes, 13 us instead of that front end) */
    int const words_before = (engine->tiny_runes_valid && engine->tiny_runes_q_count == q_count && engine->tiny_runes_c_count == c_count) ||
                             (engine->tiny_valid && engine->tiny_q_count == q_count && engine->tiny_c_count == c_count);
    if (words_before && engine->runes_needed && engine->is_unit_cost && szs_tuning_get(szs_knob_tiny_k) != 0 &&
        szs_tuning_get(szs_knob_speculate_k) != 0 && szs_tuning_get(szs_knob_tier_k) < 0) {
        size_t const narrow_before = engine->device_narrow.capacity;
        status = szs_buffer_reserve(&engine->device_narrow, szs_memory_device_k, device, (size_t)(engine->runes_needed + engine->runes_needed / 4) + 64 + SZS_NARROW_WORKSPACE, error_message);
        if (engine->device_narrow.capacity != narrow_before) engine->narrow_zeroed = NULL; /* (a new buffer, wherever it lies) */
        if (status != sz_success_k) return status;
        status = szs_cross_tiny(call, 5, NULL, 1);
        if (status != SZS_TINY_NOT_TAKEN) return status;
    }

    char *const remote = (char *)engine->device_transcode.pointer;
    uint32_t volatile *const flags = (uint32_t volatile *)engine->pinned_transcode.pointer; /* 4 flags, then `needed` */
    szs_string_ref_t *const base = (szs_string_ref_t *)engine->device_plan_refs.pointer;
    szs_plan_summary_t volatile *const summary = (szs_plan_summary_t volatile *)engine->pinned_summary.pointer;
    uint64_t *const starts = (uint64_t *)(remote + starts_at);
    uint32_t *const counts = (uint32_t *)(remote + counts_at);
    unsigned const myers_words = SZS_MYERS_MAX_WORDS * (unsigned)(engine->is_unit_cost != 0);
    szs_plan_summary_t seen;
    szs_plan_side_t q_side, c_side;
    /* (the UTF-32 buffer may move when it grows: the sides are rebuilt from it for every round) */
#define SZS_RUNE_SIDES()                                                                                                                  \
    do {                                                                                                                                  \
        szs_plan_side_t const queries_side = {call->queries->offsets, (uint64_t)(uintptr_t)engine->device_runes.pointer, q_count,        \
                                              call->queries->kind == szs_input_u64tape_k, base, base + q_count, counts, starts};         \
        q_side = queries_side, c_side = queries_side;                                                                                    \
        if (!symmetric) {                                                                                                                 \
            szs_plan_side_t const other = {call->candidates->offsets, (uint64_t)(uintptr_t)engine->device_runes.pointer, c_count,        \
                                           call->candidates->kind == szs_input_u64tape_k, base + 2 * (size_t)q_count,                    \
                                           base + 2 * (size_t)q_count + c_count, counts + q_count, starts + q_count};                    \
            c_side = other;                                                                                                               \
        }                                                                                                                                 \
    } while (0)

    /* ---- speculate (round 3): a stream of batches of one shape - the same counts, the same strings per launch width, no longer
     * longest strings, no more runes than the buffer holds, no more distinct ones than the tables have rows - is transcoded,
     * renumbered, planned AND scored without the host waiting in between: the launches of the previous call go in right behind
     * the planner, which blanks every ref if this batch does not fit them (hip/planner.hip; the byte engines' speculation, with
     * the two conditions only the device can check added to the expectation).  4096 x 4096 words of prose: the planning half
     * was as long as the scoring (profiles/r03/real_text.jsonl). */
    int const knobs_automatic = szs_tuning_get(szs_knob_speculate_k) != 0 && szs_tuning_get(szs_knob_tier_k) < 0 &&
                                szs_tuning_get(szs_knob_swap_k) < 0 && szs_tuning_get(szs_knob_cells_k) < 0 && szs_tuning_get(szs_knob_packed_k) < 0 &&
                                szs_tuning_get(szs_knob_team_k) < 0 && szs_tuning_get(szs_knob_rune_ids_k) < 0 &&
                                szs_tuning_get(szs_knob_queue_k) < 0; /* (a pinned `queue` knob makes one-group calls queue launches: those are
                                                                         planned and waited for, like the byte path's) */
    if (remembered->valid && remembered->runes && remembered->tier == SZS_TIER_LANES && remembered->use_myers && remembered->q_count == q_count &&
        remembered->c_count == c_count && remembered->symmetric == symmetric && knobs_automatic && (!remembered->alphabet || renumber) &&
        szs_decision_is_one_launch(remembered) /* see cross_device_planned */) {
        szs_decision_t const *d = remembered;
        SZS_RUNE_SIDES();
        szs_plan_expectation_t expected;
        memset(&expected, 0, sizeof(expected));
        expected.enabled = 1, expected.query_side = (uint32_t)d->transposed;
        expected.longest[0] = d->longest[0], expected.longest[1] = d->longest[1];
        memcpy(expected.variant_counts, d->variant_counts, sizeof(expected.variant_counts));
        expected.sequence = ++engine->plan_sequence;
        expected.runes_needed = (uint64_t const *)(remote + needed_at), expected.runes_capacity = engine->device_runes.capacity / sizeof(uint32_t);
        expected.alphabet_flags = (uint32_t const *)(remote + flags_at), expected.alphabet = d->alphabet;
        status = szs_call_prepare(engine, d, device, stream, error_message); /* buffers of the previous call: nothing to allocate */
        if (status != sz_success_k) return status;
        szs_call_phase(call, 2);
        hipError_t error = enqueue_transcoding(call, remote, flags_at, needed_at, staging_bytes, starts, counts, renumber);
        if (error == hipSuccess)
            error = (hipError_t)szs_hip_plan(&q_side, symmetric ? NULL : &c_side, myers_words, &expected, (szs_plan_summary_t *)summary, SZS_PLAN_VERDICTS(engine), stream);
        if (error == hipSuccess) error = hipMemcpyAsync((void *)flags, remote + flags_at, staging_bytes - flags_at, hipMemcpyDeviceToHost, stream);
        if (error != hipSuccess) {
            (void)hipStreamSynchronize(stream);
            return szs_report_hip(error, error_message); /* no scoring launch has been enqueued */
        }
        uint32_t launches = 0, cell_bits = 0;
        error = hipEventRecord(engine->event_start, stream);
        szs_string_ref_t const *const query_refs = d->transposed ? c_side.descending : q_side.descending;
        szs_string_ref_t const *const candidate_refs = d->transposed ? q_side.ascending : c_side.ascending;
        sz_status_t enqueue_status = sz_success_k;
        if (error == hipSuccess)
            error = szs_call_enqueue(engine, d, device, query_refs, candidate_refs, call->device_results, call->device_stride, stream, NULL, &launches,
                            &cell_bits, &enqueue_status, error_message);
        int stalled = 0;
        szs_decision_t scored = *d;
        engine->last_profile.planner = 2;
        status = szs_call_finish(call, &scored, error, enqueue_status, launches, cell_bits, 0, 0, &stalled);
        if (status != sz_success_k) return status;
        memcpy(&seen, (void const *)summary, sizeof(seen));
        if (seen.sequence == expected.sequence && !seen.status && seen.speculation_held) {
            szs_rocm_call_profile_t *profile = &engine->last_profile;
            uint64_t const pairs = profile->pairs;
            profile->cells = symmetric ? seen.symmetric_cells : seen.side[0].symbols * seen.side[1].symbols;
            profile->algorithmic_bytes = (symmetric ? ((uint64_t)q_count + 1) * seen.side[0].symbols
                                                    : (uint64_t)c_count * seen.side[0].symbols + (uint64_t)q_count * seen.side[1].symbols) + pairs * 16;
            profile->unique_bytes += seen.side[0].symbols + (symmetric ? 0 : seen.side[1].symbols);
            profile->longest_query = seen.side[0].longest, profile->longest_candidate = seen.side[1].longest;
            engine->runes_needed = *(uint64_t const volatile *)(flags + 4);
            remembered->summary = seen;
            remembered->plan.cells = profile->cells;
            /* words scored on the shape of an earlier batch (sentences before them, or words the tiny-token launch was not tried on): the
             * next call of these counts goes to that launch (cross_tiny) */
            if (flags[0] && szs_tiny_shaped(engine, symmetric, &seen.side[0], &seen.side[1]) && !szs_tiny_recently_refused(engine, q_count, c_count, 1)) {
                size_t const narrow_before = engine->device_narrow.capacity;
                if (szs_buffer_reserve(&engine->device_narrow, szs_memory_device_k, device, (size_t)(engine->runes_needed + engine->runes_needed / 4) + 64 + SZS_NARROW_WORKSPACE, NULL) == sz_success_k)
                    engine->tiny_runes_valid = 1, engine->tiny_runes_q_count = q_count, engine->tiny_runes_c_count = c_count;
                if (engine->device_narrow.capacity != narrow_before) engine->narrow_zeroed = NULL;
            }
            return szs_report(sz_success_k, error_message, NULL);
        }
        /* the batch has another shape, more runes or a richer alphabet: every ref was blanked, nothing real was scored */
    }

    for (int round = 0;; ++round) {
        uint64_t const capacity = engine->device_runes.capacity / sizeof(uint32_t);
        hipError_t error = enqueue_transcoding(call, remote, flags_at, needed_at, staging_bytes, starts, counts, renumber);
        SZS_RUNE_SIDES();
        szs_plan_expectation_t none;
        memset(&none, 0, sizeof(none));
        none.sequence = ++engine->plan_sequence;
        if (error == hipSuccess)
            error = (hipError_t)szs_hip_plan(&q_side, symmetric ? NULL : &c_side, myers_words, &none, (szs_plan_summary_t *)summary, SZS_PLAN_VERDICTS(engine), stream);
        if (error == hipSuccess) error = hipMemcpyAsync((void *)flags, remote + flags_at, staging_bytes - flags_at, hipMemcpyDeviceToHost, stream);
        hipError_t const drained = hipStreamSynchronize(stream); /* THE wait of the planning half; also on failure */
        if (error == hipSuccess) error = drained;
        if (error != hipSuccess) return szs_report_hip(error, error_message);
        memcpy(&seen, (void const *)summary, sizeof(seen));
        if (seen.sequence != none.sequence) return szs_report(sz_status_unknown_k, error_message, "The device planner did not report");
        if (seen.status & SZS_PLAN_STATUS_DESCENDING) return szs_report(sz_unexpected_dimensions_k, error_message, "Tape offsets must ascend");
        if (seen.status & SZS_PLAN_STATUS_OVERFLOW) return szs_report(sz_overflow_risk_k, error_message, NULL);
        uint64_t const needed = *(uint64_t const volatile *)(flags + 4);
        engine->runes_needed = needed;
        if (needed <= capacity) break;
        if (round) return szs_report(sz_status_unknown_k, error_message, "The UTF-32 buffer did not settle");
        status = szs_buffer_reserve(&engine->device_runes, szs_memory_device_k, device, (size_t)(needed + needed / 4 + 4) * sizeof(uint32_t), error_message);
        if (status != sz_success_k) return status; /* grown: transcode again, every string fits now */
    }
#undef SZS_RUNE_SIDES
    if (remembered->runes) remembered->valid = 0; /* whatever happens below, the next call is not launched on an older codepoint shape */
    if (!flags[0]) return SZS_RUNES_ARE_BYTES;
    if (seen.status & SZS_PLAN_STATUS_UNSORTED) return SZS_NOT_DEVICE_PLANNABLE; /* strings beyond the planner's histogram */
    if (szs_tiny_shaped(engine, symmetric, &seen.side[0], &seen.side[1]) && !szs_tiny_recently_refused(engine, q_count, c_count, 1)) {
        /* the summary (in RUNES) says words: their launch instead, and the next call of these counts goes there unplanned.  The narrow
         * strings get a buffer of their own - should that launch refuse the batch, the UTF-32 arrays are scored below.  `needed`
         * counts every string's BYTE span rounded up (hip/utf8.hip: transcode_tape_t::span): it bounds the bytes of both tapes. */
        size_t const narrow_before = engine->device_narrow.capacity;
        status = szs_buffer_reserve(&engine->device_narrow, szs_memory_device_k, device, (size_t)(engine->runes_needed + engine->runes_needed / 4) + 64 + SZS_NARROW_WORKSPACE, error_message);
        if (engine->device_narrow.capacity != narrow_before) engine->narrow_zeroed = NULL; /* (a new buffer, wherever it lies) */
        if (status != sz_success_k) return status;
        status = szs_cross_tiny(call, 1, &seen, 1);
        if (status != SZS_TINY_NOT_TAKEN) return status;
    }
    uint32_t const distinct = flags[1], overflowed = flags[2];
    /* the arrays hold ids 1 ... distinct: the kernels index direct tables with them */
    uint32_t const alphabet = renumber && distinct && distinct <= SZS_ALPHABET_MOST && !overflowed ? alphabet_with_room(distinct) : 0;
    szs_call_phase(call, 1);

    for (int attempt = 0; attempt < 2; ++attempt) { /* second round: a stalled band chain is re-run on the lanes tier */
        szs_decision_t d;
        uint64_t const cells = symmetric ? seen.symmetric_cells : seen.side[0].symbols * seen.side[1].symbols;
        status = szs_call_decide(engine, symmetric, 1, attempt > 0, &seen.side[0], &seen.side[1], seen.variant_counts[0], seen.variant_counts[1],
                        seen.rank_lengths, cells, &d, error_message);
        if (status != sz_success_k) return status;
        d.alphabet = alphabet;
        szs_call_decide_queue(engine, &d, seen.rank_lengths);
        status = szs_call_prepare(engine, &d, device, stream, error_message);
        if (status != sz_success_k) return status;
        szs_call_phase(call, 2);
        uint32_t launches = 0, cell_bits = 0;
        hipError_t error = hipEventRecord(engine->event_start, stream);
        szs_string_ref_t const *const query_refs = d.transposed ? c_side.descending : q_side.descending;
        szs_string_ref_t const *const candidate_refs = d.transposed ? q_side.ascending : c_side.ascending;
        sz_status_t enqueue_status = sz_success_k;
        if (error == hipSuccess)
            error = szs_call_enqueue(engine, &d, device, query_refs, candidate_refs, call->device_results, call->device_stride, stream, NULL, &launches,
                            &cell_bits, &enqueue_status, error_message);
        int stalled = 0;
        engine->last_profile.planner = 1;
        status = szs_call_finish(call, &d, error, enqueue_status, launches, cell_bits, seen.side[0].symbols, seen.side[1].symbols, &stalled);
        if (status != sz_success_k) return status;
        if (!stalled) {
            *remembered = d; /* the next batch of this shape goes in behind its own planner, unseen by the host */
            remembered->summary = seen, remembered->refs_current = 0;
            return sz_success_k;
        }
    }
    return szs_report(sz_status_unknown_k, error_message, "Systolic pipeline stalled");
}

