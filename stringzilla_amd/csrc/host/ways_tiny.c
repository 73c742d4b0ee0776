/*
 *  ways_tiny.c - the tiny-token regime of one engine call (split from dispatch.c in round 6; see dispatch_internal.h).
 */
#include "dispatch_internal.h"

#include <stdio.h>
#include <stdlib.h>
#include <string.h>

/* ---- the tiny-token regime (hip/myers_tiny.hip; reference: cuda.cuh:2864, :4297-4340) ------------------------------------- */


/** Tiny tokens on both sides, enough of them to fill the device - or whatever the `tiny` knob says. */
int szs_tiny_shaped(szs_engine_s const *engine, int symmetric, szs_side_stats_t const *queries, szs_side_stats_t const *candidates) {
    int const knob = szs_tuning_get(szs_knob_tiny_k);
    if (!engine->is_unit_cost || knob == 0) return 0;
    if (symmetric) candidates = queries; /* (round 6: one tape against itself - the launch scores the whole square, both triangles) */
    if (engine->family != szs_family_levenshtein_k && engine->family != szs_family_levenshtein_utf8_k /* an ASCII corpus */) return 0;
    /* a string beyond that launch's 255 bytes (an occasional long line among the words: the planner's summary knows the longest) - the
     * launch would refuse the call after scoring most of it, every call again */
    if (queries->longest > SZS_TINY_LONGEST || candidates->longest > SZS_TINY_LONGEST) return 0;
    if (knob > 0) return 1;
    /* word-like: mean length well under the sixteen rows of that kernel's bit-vectors (what is longer - a few per cent of a text's
     * tokens - rides along in the same launch; the kernel itself says when a string is beyond it) and a matrix worth a launch */
    return queries->symbols <= 10ull * queries->count && candidates->symbols <= 10ull * candidates->count && queries->count >= 64 &&
           candidates->count >= 1024 && (uint64_t)queries->count * candidates->count >= (1ull << 20);
}

/** The tiny-token kernel refused a recent batch of these counts (cross_tiny): the next sixteen such calls do not try it again. */
int szs_tiny_recently_refused(szs_engine_s *engine, uint32_t q_count, uint32_t c_count, int count_down) {
    if (engine->tiny_refused <= 0 || engine->tiny_q_count != q_count || engine->tiny_c_count != c_count) return 0;
    if (szs_tuning_get(szs_knob_tiny_k) >= 0) return 0; /* a pinned knob is obeyed every time */
    if (count_down) --engine->tiny_refused;
    return 1;
}

/**
 *  One launch of the tiny-token kernel, straight from the caller's tapes, and the call's wait.  sz_success_k: scored.
 *  SZS_TINY_NOT_TAKEN: the kernel met a string beyond 255 bytes or malformed offsets - nothing it wrote counts, the caller goes on
 *  to the ordinary path (which also reports malformed tapes).  `planner_mode`: 1 when a planner's summary chose this kernel, 5 when
 *  the previous call of the engine did and nothing was planned at all.
 */
sz_status_t szs_cross_tiny(szs_call_t *call, uint32_t planner_mode, szs_plan_summary_t const *seen /* or NULL */, int runes) {
    szs_engine_s *engine = call->engine;
    hipStream_t const stream = call->stream;
    uint32_t volatile *const unfit = (uint32_t volatile *)((char *)engine->pinned_summary.pointer + 992);
    unsigned long long volatile *const symbols = (unsigned long long volatile *)((char *)engine->pinned_summary.pointer + 976);
    if (!++engine->plan_sequence) ++engine->plan_sequence;
    uint32_t const sequence = engine->plan_sequence;
    *unfit = 0, symbols[0] = symbols[1] = 0;
    szs_call_phase(call, 2);
    szs_tape_t q_tape = {call->queries->offsets, (uint64_t)(uintptr_t)call->queries->data, call->q_count, call->queries->kind == szs_input_u64tape_k};
    szs_tape_t c_tape = q_tape; /* a symmetric call: the one tape against itself, the whole square (what the ordinary path leaves too) */
    if (!call->symmetric) {
        szs_tape_t const other = {call->candidates->offsets, (uint64_t)(uintptr_t)call->candidates->data, call->c_count,
                                  call->candidates->kind == szs_input_u64tape_k};
        c_tape = other;
    }
    uint32_t launches = 0;
    /* ... and the cells of its lower triangle are ((sum len)^2 + sum len^2) / 2: the launch leaves a partial sum of squares per block of
     * 256 strings in pinned memory */
    size_t const square_blocks = call->symmetric ? ((size_t)call->c_count + 255) / 256 : 0;
    uint64_t volatile *squares = NULL;
    if (square_blocks) {
        sz_status_t const reserved = szs_buffer_reserve(&engine->pinned_squares, szs_memory_pinned_k, call->device, square_blocks * sizeof(uint64_t), call->error_message);
        if (reserved != sz_success_k) return reserved;
        squares = (uint64_t volatile *)engine->pinned_squares.pointer;
        for (size_t b = 0; b < square_blocks; ++b) squares[b] = 0;
    }
    /* The codepoint engine (round 6; reference: cuda.cuh:3294): one pass ahead of the launch turns both UTF-8 tapes into byte strings of
     * rune ids (hip/utf8.hip: utf8_narrow_kernel) in a buffer of the engine's, and the launch scores THOSE - no transcoding to UTF-32,
     * no renumbering passes, no planner.  Device staging (cross_device_planned_runes reserved it): a word per string. */
    uint64_t *const entries = runes ? (uint64_t *)engine->device_transcode.pointer : NULL;
    /* the head of the narrow buffer: the pass's table of claimed runes - it LIVES ON from call to call, zeroed when the buffer is new
     * and after a batch that the pass refused (a full table, perhaps) - and the sides' totals of runes */
    char *const narrow_workspace = runes ? (char *)engine->device_narrow.pointer : NULL;
    char *const narrow_strings = runes ? narrow_workspace + SZS_NARROW_WORKSPACE : NULL;
    uint64_t *trace = NULL; /* `trace` knob: the phases of every workgroup of the launch (printed after the wait) */
    size_t const trace_workgroups = 8192, trace_slots = 10;
    hipError_t error = hipEventRecord(engine->event_start, stream);
    if (call->trace && szs_buffer_reserve(&engine->device_queue_trace, szs_memory_device_k, call->device, trace_workgroups * trace_slots * 8, NULL) == sz_success_k) {
        trace = (uint64_t *)engine->device_queue_trace.pointer;
        if (hipMemsetAsync(trace, 0, trace_workgroups * trace_slots * 8, stream) != hipSuccess) trace = NULL;
    }
    /* ONE launch: the tiny tokens and, in their shadow, the few longer ones (hip/myers_tiny.hip).  (Round 5's first design was four:
     * a pass that listed the longer strings and tabled the tiny ones' masks in device memory, the outliers' kernel, a pass that set
     * the tables back, the tiny-token kernel - 100 us of kernels on 4096 x 4096 words of text where the one launch takes 81.) */
    if (error == hipSuccess && runes) {
        if (engine->narrow_zeroed != (void *)narrow_workspace) {
            error = hipMemsetAsync(narrow_workspace, 0, SZS_NARROW_WORKSPACE, stream);
            engine->narrow_zeroed = error == hipSuccess ? (void *)narrow_workspace : NULL;
        }
        szs_tape_t narrowed_too = c_tape;
        if (call->symmetric) narrowed_too.count = 0; /* (the one tape is narrowed once) */
        if (error == hipSuccess)
            error = (hipError_t)szs_hip_utf8_narrow(&q_tape, &narrowed_too, narrow_strings, engine->device_narrow.capacity - SZS_NARROW_WORKSPACE, entries,
                                                    narrow_workspace, (uint32_t *)unfit, sequence, stream);
        launches += error == hipSuccess;
        szs_tape_t const q_narrow = {entries, (uint64_t)(uintptr_t)narrow_strings, call->q_count, 2};
        szs_tape_t const c_narrow = {entries + call->q_count, (uint64_t)(uintptr_t)narrow_strings, call->c_count, 2};
        q_tape = q_narrow, c_tape = call->symmetric ? q_narrow : c_narrow;
    }
    if (error == hipSuccess) {
        error = (hipError_t)szs_hip_levenshtein_tiny(&q_tape, &c_tape, (uint64_t *)call->device_results, call->device_stride, (uint32_t *)unfit, sequence,
                                                           (unsigned long long *)symbols,
                                                           runes ? (uint64_t *)(narrow_workspace + SZS_NARROW_SLOTS * sizeof(uint32_t)) : NULL, (uint64_t *)squares, trace,
                                                           trace_workgroups, szs_tuning_get(szs_knob_tiny_k) == 2, stream);
        launches += error == hipSuccess;
    }
    engine->last_streams = 1;
    /* what szs_call_finish() reads: lanes tier, one launch.  On the stack, like the other paths' copies of a decision: a failed allocation here
     * would have returned with the launch still writing the caller's matrix and the pinned words */
    szs_decision_t shape_of_call;
    memset(&shape_of_call, 0, sizeof(shape_of_call));
    szs_decision_t *const shape = &shape_of_call;
    shape->tier = SZS_TIER_LANES, shape->q_count = call->q_count, shape->c_count = call->c_count, shape->runes = runes;
    if (seen) shape->longest[0] = seen->side[0].longest, shape->longest[1] = seen->side[1].longest;
    engine->last_profile.planner = planner_mode;
    int stalled = 0;
    sz_status_t status = szs_call_finish(call, shape, error, sz_success_k, launches, 0, 0, 0, &stalled);
    if (runes && (status != sz_success_k || *unfit == sequence)) engine->narrow_zeroed = NULL; /* a full table, totals nobody read: start over */
    if (status != sz_success_k) return status;
    if (trace) { /* where a workgroup of the tiny-token kernel spends its time: mean ticks (10 ns) between its stamps */
        size_t const slots = trace_slots, last_slot = 8;
        uint64_t *const ticks = (uint64_t *)malloc(trace_workgroups * slots * 8);
        if (ticks && hipMemcpy(ticks, trace, trace_workgroups * slots * 8, hipMemcpyDeviceToHost) == hipSuccess) {
            double sums[10] = {0};
            uint64_t first = ~0ull, last = 0;
            size_t seen = 0;
            for (size_t w = 0; w < trace_workgroups; ++w) {
                if (!ticks[slots * w] || !ticks[slots * w + last_slot]) continue;
                ++seen;
                first = ticks[slots * w] < first ? ticks[slots * w] : first, last = ticks[slots * w + last_slot] > last ? ticks[slots * w + last_slot] : last;
                for (size_t k = 1; k < slots; ++k)
                    if (ticks[slots * w + k] && ticks[slots * w + k - 1]) sums[k] += (double)(ticks[slots * w + k] - ticks[slots * w + k - 1]);
            }
            uint64_t last_begin = 0, longest_life = 0, first_end = ~0ull;
            double lives = 0;
            for (size_t w = 0; w < trace_workgroups; ++w) {
                if (!ticks[slots * w] || !ticks[slots * w + last_slot]) continue;
                uint64_t const life = ticks[slots * w + last_slot] - ticks[slots * w];
                last_begin = ticks[slots * w] > last_begin ? ticks[slots * w] : last_begin, longest_life = life > longest_life ? life : longest_life, lives += (double)life;
                first_end = ticks[slots * w + last_slot] < first_end ? ticks[slots * w + last_slot] : first_end;
            }
            if (seen)
                fprintf(stderr, "tiny kernel: last begin at %.1f us, first end at %.1f us; a workgroup lives %.1f us on average, %.1f at most\n", (last_begin - first) * 1e-2,
                        (first_end - first) * 1e-2, lives / seen * 1e-2, longest_life * 1e-2);
            if (seen)
                fprintf(stderr, "tiny kernel: %zu workgroups over %.1f us; mean us per workgroup: offsets + local sort %.2f, texts %.2f, first masks %.2f, columns %.2f, "
                                "%s %.2f, un-build + stores %.2f, rest (further groups) %.2f, long queries %.2f\n", seen, (last - first) * 1e-2, sums[1] / seen * 1e-2,
                        sums[2] / seen * 1e-2, sums[3] / seen * 1e-2, sums[4] / seen * 1e-2, "long candidates + barrier", sums[5] / seen * 1e-2,
                        sums[6] / seen * 1e-2, sums[7] / seen * 1e-2, sums[8] / seen * 1e-2);
        }
        free(ticks);
    }
    if (*unfit == sequence) {
        /* refused (a block or span too dense in long strings, a string beyond 255 bytes): remember the counts, so that a stream of
         * such batches does not pay this launch and its wait on every call because their summaries look like words */
        if (runes) engine->tiny_runes_valid = 0;
        else engine->tiny_valid = 0;
        if (planner_mode == 1) { /* ... a batch whose SUMMARY looked like words (clustered long lines among short ones).  A batch that
                                    came straight here on the previous call's word (mode 5: sentences after words) is judged by its
                                    own summary next time - nothing to remember */
            engine->tiny_refused = 16; /* calls of these counts that go straight to the ordinary path */
            engine->tiny_q_count = call->q_count, engine->tiny_c_count = call->c_count;
        }
        return SZS_TINY_NOT_TAKEN;
    }
    engine->tiny_refused = 0;
    szs_rocm_call_profile_t *profile = &engine->last_profile;
    uint64_t const q_symbols = symbols[0], c_symbols = symbols[1];
    profile->cells = q_symbols * c_symbols;
    profile->algorithmic_bytes = (uint64_t)call->c_count * q_symbols + (uint64_t)call->q_count * c_symbols + profile->pairs * 16;
    profile->unique_bytes += q_symbols + c_symbols;
    if (call->symmetric) { /* (the conventions of complete_from_summary) */
        uint64_t sum_of_squares = 0;
        for (size_t b = 0; b < square_blocks; ++b) sum_of_squares += squares[b];
        profile->cells = (q_symbols * q_symbols + sum_of_squares) / 2;
        profile->algorithmic_bytes = ((uint64_t)call->q_count + 1) * q_symbols + profile->pairs * 16;
        profile->unique_bytes -= c_symbols;
    }
    /* the next call of these counts comes straight here - as long as the batch keeps looking like tiny tokens */
    szs_side_stats_t now[2];
    memset(now, 0, sizeof(now));
    now[0].count = call->q_count, now[0].symbols = q_symbols, now[1].count = call->c_count, now[1].symbols = c_symbols;
    if (runes) engine->tiny_runes_valid = szs_tiny_shaped(engine, call->symmetric, &now[0], &now[1]), engine->tiny_runes_q_count = call->q_count, engine->tiny_runes_c_count = call->c_count;
    else engine->tiny_valid = szs_tiny_shaped(engine, call->symmetric, &now[0], &now[1]), engine->tiny_q_count = call->q_count, engine->tiny_c_count = call->c_count;
    if (engine->remembered) engine->remembered->refs_current = 0, engine->remembered->valid = 0; /* another kind of call came between */
    return szs_report(sz_success_k, call->error_message, NULL);
}

