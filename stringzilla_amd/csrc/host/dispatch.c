/*
 *  dispatch.c - one engine call, start to finish: normalise inputs -> plan -> decide -> launch -> synchronise.
 *
 *  ROCm counterpart of the reference's `cross_()` / `run_trampoline_()` (cuda.cuh:4247-4417,4435-4741) and
 *  `cuda_weighted_cross_()` (cuda.cuh:5913).  Same observable behaviour - synchronous call, results in the caller's
 *  matrix, device-accessibility checks on the strings, staged copy when `results` is not device-visible - with a
 *  different mechanism (see plan.c, hip/planner.hip).  Nothing here computes a score on the CPU.
 *
 *  Two ways to the same launches:
 *    device-planned  both inputs are tapes whose offsets the GPU can read: hip/planner.hip turns the offsets into sorted refs
 *                    and a summary; the host never reads an offset.  When the previous call of this engine had the same
 *                    counts, the scoring launches are enqueued BEHIND the planner right away, shaped like that call; the
 *                    planner blanks the refs if the batch does not fit that shape and the host, after the call's one
 *                    synchronisation, re-plans from the summary.  A batch stream of stable shape thus costs one planner
 *                    launch + the scoring launches + one wait - no download, no host sort, no upload.
 *    host-planned    callback sequences, the codepoint engine (needs the transcoding pass first), offsets only the host
 *                    can read, strings too long for the device planner's histogram: as in round 1, minus the per-call
 *                    allocations.
 *  Every failure after the first enqueue leaves through one exit that drains the stream first: the call is synchronous
 *  also when it fails, so the caller may free its buffers the moment it returns.
 */
#include "dispatch_internal.h"

#include <dlfcn.h>
#include <pthread.h>

#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

static double now_milliseconds(void) {
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return ts.tv_sec * 1e3 + ts.tv_nsec * 1e-6;
}

szs_pointer_traits_t szs_classify_pointer(void const *pointer) {
    szs_pointer_traits_t traits = {1, 0, 0};
    if (!pointer) return traits;
    hipPointerAttribute_t attributes;
    memset(&attributes, 0, sizeof(attributes));
    hipError_t const error = hipPointerGetAttributes(&attributes, pointer);
    if (error != hipSuccess) { /* older runtimes report plain host memory as an error */
        (void)hipGetLastError();
        return traits;
    }
    switch (attributes.type) {
    case hipMemoryTypeDevice: traits.host_readable = 0, traits.device_accessible = 1, traits.device_resident = 1; break;
    case hipMemoryTypeHost: traits.host_readable = 1, traits.device_accessible = 1; break;
    case hipMemoryTypeManaged: traits.host_readable = 1, traits.device_accessible = 1; break;
    default: traits.host_readable = 1, traits.device_accessible = 0; break; /* unregistered host memory */
    }
    return traits;
}

/**
 *  Tapes whose offsets live in device-only memory must be read by the host planner: this ENQUEUES their download into
 *  the pinned staging area (no synchronisation - the caller waits once for both sides) and returns where the host will
 *  find the offsets; host-readable offsets are returned as they are.  `*pending` is set when a copy was enqueued.
 */
sz_status_t szs_prefetch_offsets(void *pinned_staging, hipStream_t stream, szs_input_t const *input, size_t staging_offset,
                                 void const **host_offsets, int *pending, char const **error_message) {
    *host_offsets = input->offsets;
    if (input->kind == szs_input_sequence_k) return sz_success_k;
    if (!input->offsets) return szs_report(sz_status_unknown_k, error_message, "Tape offsets must not be null");
    if (szs_classify_pointer(input->offsets).host_readable) return sz_success_k;
    size_t const offset_size = input->kind == szs_input_u32tape_k ? 4 : 8;
    void *landing = (char *)pinned_staging + staging_offset;
    hipError_t const error = hipMemcpyAsync(landing, input->offsets, (input->count + 1) * offset_size,
                                            hipMemcpyDeviceToHost, stream);
    if (error != hipSuccess) return szs_report_hip(error, error_message);
    *host_offsets = landing, *pending = 1;
    return sz_success_k;
}

/**
 *  Produces absolute addresses and 32-bit lengths for every string of one side, from a callback sequence or from a
 *  tape whose offsets are readable at `offsets` (see prefetch_offsets).
 */
sz_status_t szs_gather_strings(szs_input_t const *input, void const *offsets, uint64_t *addresses, uint32_t *lengths,
                               uint64_t *total_bytes, int *needs_staging, char const **error_message) {
    size_t const count = input->count;
    *total_bytes = 0;
    if (needs_staging) *needs_staging = 0;
    if (input->kind == szs_input_sequence_k) {
        sz_sequence_t const *sequence = input->sequence;
        int checked = 0;
        for (size_t i = 0; i < count; ++i) {
            char const *start = sequence->get_start(sequence->handle, i);
            size_t const length = sequence->get_length(sequence->handle, i);
            if (length > 0xFFFFFFFFull) return szs_report(sz_overflow_risk_k, error_message, NULL);
            if (length && !checked) { /* like the reference, vet one representative string (cuda.cuh:4268-4272) */
                if (!szs_classify_pointer(start).device_accessible) {
                    if (!needs_staging) return szs_report(sz_device_memory_mismatch_k, error_message, NULL);
                    *needs_staging = 1;
                }
                checked = 1;
            }
            addresses[i] = (uint64_t)(uintptr_t)start, lengths[i] = (uint32_t)length;
            *total_bytes += length;
        }
        return sz_success_k;
    }

    size_t const offset_size = input->kind == szs_input_u32tape_k ? 4 : 8;
    uint64_t const base = (uint64_t)(uintptr_t)input->data;
    if (offset_size == 4) {
        uint32_t const *o = (uint32_t const *)offsets;
        for (size_t i = 0; i < count; ++i) {
            if (o[i + 1] < o[i]) return szs_report(sz_unexpected_dimensions_k, error_message, "Tape offsets must ascend");
            addresses[i] = base + o[i], lengths[i] = o[i + 1] - o[i];
        }
        *total_bytes = (uint64_t)o[count] - o[0];
    }
    else {
        uint64_t const *o = (uint64_t const *)offsets;
        for (size_t i = 0; i < count; ++i) {
            if (o[i + 1] < o[i]) return szs_report(sz_unexpected_dimensions_k, error_message, "Tape offsets must ascend");
            uint64_t const length = o[i + 1] - o[i];
            if (length > 0xFFFFFFFFull) return szs_report(sz_overflow_risk_k, error_message, NULL);
            addresses[i] = base + o[i], lengths[i] = (uint32_t)length;
        }
        *total_bytes = o[count] - o[0];
    }
    if (*total_bytes && !szs_classify_pointer(input->data).device_accessible) {
        if (!needs_staging) return szs_report(sz_device_memory_mismatch_k, error_message, NULL);
        *needs_staging = 1;
    }
    return sz_success_k;
}

/**
 *  `cpu_requests = gpu` only (host/tuning.c): a caller that asked for a CPU engine hands over strings in plain host
 *  memory, which no kernel can read.  Their bytes are packed into the engine's pinned staging tape, shipped to HBM in one
 *  copy, and `addresses` are rewritten to point there.  (By default such inputs are refused with
 *  sz_device_memory_mismatch_k, like the reference's GPU engines do: cuda.cuh:4268-4272.)
 */
static sz_status_t stage_host_strings(szs_engine_s *engine, hipStream_t stream, uint64_t *addresses, uint32_t const *lengths,
                                      uint32_t count, uint64_t total_bytes, size_t tape_offset, char const **error_message) {
    if (!total_bytes) return sz_success_k;
    char *const pinned = (char *)engine->pinned_tape.pointer + tape_offset, *const remote = (char *)engine->device_tape.pointer + tape_offset;
    uint64_t cursor = 0;
    for (uint32_t i = 0; i < count; ++i) {
        memcpy(pinned + cursor, (void const *)(uintptr_t)addresses[i], lengths[i]);
        addresses[i] = (uint64_t)(uintptr_t)(remote + cursor);
        cursor += lengths[i];
    }
    hipError_t const error = hipMemcpyAsync(remote, pinned, total_bytes, hipMemcpyHostToDevice, stream);
    if (error != hipSuccess) return szs_report_hip(error, error_message);
    return sz_success_k;
}

/**
 *  Transcodes both sides (one side when symmetric) to UTF-32 on the device and, if the corpus is not pure ASCII,
 *  rewrites `addresses` / `lengths` in place to point at the runes and to count runes.  One extra synchronisation:
 *  the planner needs the rune counts.  String i's runes start at the prefix sum of BYTE lengths (runes <= bytes).
 */
static sz_status_t transcode_to_runes(szs_engine_s *engine, int device, hipStream_t stream, int symmetric,
                                      uint64_t *q_addresses, uint32_t *q_lengths, uint32_t q_count,
                                      uint64_t *c_addresses, uint32_t *c_lengths, uint32_t c_count, int *runes,
                                      uint32_t *alphabet, char const **error_message) {
    size_t const strings = (size_t)q_count + (symmetric ? 0 : c_count);
    /* staging layout, host and device alike: [refs][rune starts][rune counts][flag][distinct runes][alphabet overflow] */
    size_t const refs_at = 0, starts_at = refs_at + strings * sizeof(szs_string_ref_t);
    size_t const counts_at = starts_at + strings * sizeof(uint64_t), flag_at = counts_at + strings * sizeof(uint32_t);
    size_t const staging_bytes = flag_at + 3 * sizeof(uint32_t);
    *alphabet = 0;
    sz_status_t status = szs_buffer_reserve(&engine->pinned_transcode, szs_memory_pinned_k, device, staging_bytes, error_message);
    if (status != sz_success_k) return status;
    status = szs_buffer_reserve(&engine->device_transcode, szs_memory_device_k, device, staging_bytes, error_message);
    if (status != sz_success_k) return status;

    char *const host = (char *)engine->pinned_transcode.pointer, *const remote = (char *)engine->device_transcode.pointer;
    szs_string_ref_t *refs = (szs_string_ref_t *)(host + refs_at);
    uint64_t *starts = (uint64_t *)(host + starts_at);
    uint64_t total = 0;
    for (size_t i = 0; i < strings; ++i) {
        int const is_query = i < q_count;
        size_t const k = is_query ? i : i - q_count;
        refs[i].address = is_query ? q_addresses[k] : c_addresses[k];
        refs[i].length = is_query ? q_lengths[k] : c_lengths[k];
        refs[i].index = (uint32_t)i;
        starts[i] = total, total += (refs[i].length + 3u) & ~(uint64_t)3; /* 16-byte aligned UTF-32 arrays: the kernels load four runes at once */
    }
    *(uint32_t *)(host + flag_at) = 0;
    status = szs_buffer_reserve(&engine->device_runes, szs_memory_device_k, device, (total + 4) * sizeof(uint32_t), error_message);
    if (status != sz_success_k) return status;
    /* A batch worth the three extra launches gets its runes renumbered 1 ... A (hip/utf8.hip: the codepoint kernels then index a
     * table instead of probing one); the `alphabet` knob: 0 never, 1 always. */
    int const alphabet_knob = szs_tuning_get(szs_knob_alphabet_k);
    int const renumber = alphabet_knob == 0 ? 0 : alphabet_knob > 0 ? 1 : total >= SZS_ALPHABET_WORTH_BYTES;
    if (renumber) {
        status = szs_buffer_reserve(&engine->device_alphabet, szs_memory_device_k, device, szs_hip_alphabet_workspace_bytes(), error_message);
        if (status != sz_success_k) return status;
    }

    hipError_t error = hipMemcpyAsync(remote, host, counts_at, hipMemcpyHostToDevice, stream);
    if (error == hipSuccess) error = hipMemsetAsync(remote + flag_at, 0, 3 * sizeof(uint32_t), stream);
    if (error == hipSuccess) {
        int const launch_error = szs_hip_utf8_transcode(
            (szs_string_ref_t const *)(remote + refs_at), (uint32_t)strings, (uint64_t const *)(remote + starts_at),
            (uint32_t *)engine->device_runes.pointer, (uint32_t *)(remote + counts_at), (uint32_t *)(remote + flag_at), stream);
        error = (hipError_t)launch_error;
    }
    if (error == hipSuccess && renumber)
        error = (hipError_t)szs_hip_alphabet_rename((uint32_t)strings, (uint64_t const *)(remote + starts_at), (uint32_t const *)(remote + counts_at),
                                                    (uint32_t *)engine->device_runes.pointer, (uint32_t const *)(remote + flag_at),
                                                    engine->device_alphabet.pointer, 0, SZS_ALPHABET_MOST,
                                                    (uint32_t *)(remote + flag_at + sizeof(uint32_t)), stream);
    if (error == hipSuccess)
        error = hipMemcpyAsync(host + counts_at, remote + counts_at, staging_bytes - counts_at, hipMemcpyDeviceToHost, stream);
    hipError_t const drained = hipStreamSynchronize(stream); /* also on failure: nothing stays in flight */
    if (error == hipSuccess) error = drained;
    if (error != hipSuccess) return szs_report_hip(error, error_message);

    *runes = *(uint32_t const *)(host + flag_at) != 0;
    if (!*runes) return sz_success_k;
    uint32_t const distinct = ((uint32_t const *)(host + flag_at))[1], overflowed = ((uint32_t const *)(host + flag_at))[2];
    if (renumber && distinct && distinct <= SZS_ALPHABET_MOST && !overflowed) *alphabet = distinct; /* the arrays now hold ids */
    uint32_t const *counts = (uint32_t const *)(host + counts_at);
    uint64_t const base = (uint64_t)(uintptr_t)engine->device_runes.pointer;
    for (size_t i = 0; i < strings; ++i) {
        uint64_t const address = base + starts[i] * sizeof(uint32_t);
        if (i < q_count) q_addresses[i] = address, q_lengths[i] = counts[i];
        else c_addresses[i - q_count] = address, c_lengths[i - q_count] = counts[i];
    }
    if (symmetric) {
        memcpy(c_addresses, q_addresses, (size_t)q_count * sizeof(uint64_t));
        memcpy(c_lengths, q_lengths, (size_t)q_count * sizeof(uint32_t));
    }
    return sz_success_k;
}

static void fill_cost_model(szs_engine_s const *engine, int transposed, szs_cost_model_t *model) {
    memset(model, 0, sizeof(*model));
    if (engine->family == szs_family_levenshtein_k || engine->family == szs_family_levenshtein_utf8_k) {
        /* Minimising non-negative costs == maximising their negation; the kernel negates the result back. */
        model->uniform_match = -(int32_t)engine->match, model->uniform_mismatch = -(int32_t)engine->mismatch;
        model->gap_open = -(int32_t)engine->open, model->gap_extend = -(int32_t)engine->extend;
        if (engine->uniform_classes) memcpy(model->byte_to_class, engine->uniform_byte_to_class, 256); /* the team tier's dense alphabet */
    }
    else {
        /* cost(query, candidate) = table[class(query)][class(candidate)] (serial.hpp:199-204): when the planner swapped
         * the sides, the kernel's "query" is the caller's candidate, so it must see the transposed table. */
        for (int i = 0; i < 32; ++i)
            for (int j = 0; j < 32; ++j)
                model->substitution[i * 32 + j] = transposed ? engine->class_costs[j * 32 + i] : engine->class_costs[i * 32 + j];
        memcpy(model->byte_to_class, engine->byte_to_class, 256);
        model->gap_open = engine->open, model->gap_extend = engine->extend;
    }
}

static void release_device_state(szs_engine_s *engine) {
    szs_buffer_release(&engine->pinned_staging);
    szs_buffer_release(&engine->device_refs);
    szs_buffer_release(&engine->device_results);
    szs_buffer_release(&engine->device_boundary);
    szs_buffer_release(&engine->device_model);
    szs_buffer_release(&engine->device_systolic);
    szs_buffer_release(&engine->device_tape);
    szs_buffer_release(&engine->pinned_tape);
    szs_buffer_release(&engine->device_runes);
    szs_buffer_release(&engine->device_transcode);
    szs_buffer_release(&engine->pinned_transcode);
    szs_buffer_release(&engine->device_alphabet);
    szs_buffer_release(&engine->device_narrow);
    szs_buffer_release(&engine->device_plan_refs);
    szs_buffer_release(&engine->device_presence);
    szs_buffer_release(&engine->device_queue);
    szs_buffer_release(&engine->device_queue_trace);
    engine->queue_zeroed = NULL;
    szs_buffer_release(&engine->pinned_summary);
    szs_buffer_release(&engine->pinned_squares);
    szs_buffer_release(&engine->device_fused);
    engine->tiny_valid = 0, engine->tiny_runes_valid = 0, engine->narrow_zeroed = NULL;
    engine->fused_zeroed = NULL;
    if (engine->events_device >= 0) {
        (void)hipEventDestroy(engine->event_start);
        (void)hipEventDestroy(engine->event_stop);
        engine->events_device = -1;
    }
    if (engine->aux_device >= 0) { /* the streams belong to the process-wide pool (szs_aux_streams); the events are this engine's */
        for (int i = 0; i < SZS_AUX_STREAMS; ++i) (void)hipEventDestroy(engine->aux_done[i]);
        (void)hipEventDestroy(engine->fork_event);
        engine->aux_device = -1;
    }
    engine->model_uploaded_device = -1;
    if (engine->remembered) engine->remembered->valid = 0;
}

void szs_engine_release(szs_engine_s *engine) {
    if (engine->device >= 0) {
        int previous = 0;
        (void)hipGetDevice(&previous);
        (void)hipSetDevice(engine->device);
        release_device_state(engine);
        (void)hipSetDevice(previous);
    }
    szs_buffer_release(&engine->host_lengths);
    szs_buffer_release(&engine->host_scratch);
    free(engine->remembered);
    engine->remembered = NULL;
}

/* ---- the decision: everything that follows from the statistics of the two sides ---------------------------------------- */

/**
 *  Which instance of the team tier scores a 16-bit class-table call, 0 for the one-pair-per-lane kernel: the `team` knob, or
 *  the lanes-per-item rule of plan.c with the instance compiled for that many lanes.
 */
static unsigned team_shape_for(int affine, uint32_t classes, szs_side_stats_t const *queries, szs_side_stats_t const *candidates) {
    int const knob = szs_tuning_get(szs_knob_team_k);
    if (knob == 0) return 0;
    if (knob > 0) return szs_hip_weighted_team_has_shape((unsigned)knob) && szs_hip_weighted_team_fits((unsigned)knob, classes) ? (unsigned)knob : 0;
    unsigned lanes = szs_plan_team_lanes(affine, queries, candidates);
    for (; lanes; lanes = lanes > 16 ? 16 : lanes > 4 ? 4 : 0) { /* sixty-four strips fit a CU's LDS up to ~19 classes (DNA), sixteen strips
                                                                     of a rich alphabet (text: ~95 classes) do not either: four do */
        /* Rows per lane: 16 (four wavefronts per SIMD, half the profile) when the longest query fits ONE pass of that shape
         * anyway - 4 x 16 rows: +17 ... 20 % at 24 ... 48 rows; 16 x 16: +4 ... 21 % at 192 ... 256 - or when the alphabet is
         * rich: four strips of 95 classes are 55 KB at 32 rows, two workgroups per CU (non-unit Levenshtein costs over text:
         * 12.3 -> 15.3 TCUPS at 16).  Else 32: half the passes over the candidates.  (team_sweep_v2.jsonl) */
        unsigned const short_rows = lanes == 4 ? 80u : affine ? 240u : 320u;
        unsigned const wanted_registers = queries->longest <= short_rows || (lanes == 4 && classes > 48) ? 16u : 32u;
        unsigned fallback = 0;
        for (unsigned index = 0; szs_hip_weighted_team_shape(index); ++index) {
            unsigned const shape = szs_hip_weighted_team_shape(index);
            if (shape / 10000u != lanes || !szs_hip_weighted_team_fits(shape, classes)) continue;
            if (shape / 100u % 100u == wanted_registers) return shape;
            if (!fallback) fallback = shape; /* the first compiled instance of that width */
        }
        if (fallback) return fallback;
    }
    return 0;
}

sz_status_t szs_call_decide(szs_engine_s const *engine, int symmetric, int runes, int force_lanes, szs_side_stats_t const *q_stats,
                          szs_side_stats_t const *c_stats, uint32_t const *q_variants, uint32_t const *c_variants,
                          uint32_t const (*ranks)[SZS_PLAN_RANK_SAMPLES + 1] /* the caller's sides, or NULL: not known yet */,
                          uint64_t cells, szs_decision_t *d, char const **error_message) {
    memset(d, 0, sizeof(*d));
    d->symmetric = symmetric, d->runes = runes;
    d->q_count = q_stats->count, d->c_count = c_stats->count;
    d->longest[0] = q_stats->longest, d->longest[1] = c_stats->longest;
    d->use_myers = engine->is_unit_cost && (engine->family == szs_family_levenshtein_k || engine->family == szs_family_levenshtein_utf8_k);
    d->maximise = engine->family == szs_family_needleman_wunsch_k || engine->family == szs_family_smith_waterman_k;
    /* bit-parallel at any length (2048-row strips beyond 64 words); `banded` names the byte flavour of the strip kernel */
    d->banded = d->use_myers && !runes;

    /* ---- cell width (reach rule, serial.hpp:135-162,370-386): 16-bit pairs, 32 bits, or the 64-bit tier (hip/wide.hip).
     * It follows from the longest string of either side, whichever of them ends up on the workgroups. */
    uint64_t const span = d->maximise ? (uint64_t)q_stats->longest + c_stats->longest
                                      : (q_stats->longest > c_stats->longest ? q_stats->longest : c_stats->longest);
    uint64_t const magnitude = engine->magnitude ? engine->magnitude : 1;
    uint64_t const reach = (span + (engine->is_linear ? 1 : 3)) * magnitude;
    /* Unit-cost engines never need wide cells: their distances are bounded by the longer string, below 2^32 by construction. */
    d->wide_cells = szs_tuning_get(szs_knob_cells_k) == 64 || (!d->use_myers && reach >= 0x7FFFFFF0ull);

    d->objective = engine->family == szs_family_needleman_wunsch_k   ? szs_objective_global_k
                   : engine->family == szs_family_smith_waterman_k
                       ? (engine->open <= 0 && engine->extend <= 0 ? szs_objective_local_saturating_k : szs_objective_local_k)
                   : runes                                         ? szs_objective_distance_runes_k
                                                                   : szs_objective_distance_k;
    /* 16-bit strip boundaries when every parked value provably fits: global scores are bounded by the reach, saturating
     * local ones by (shorter side) x (largest cost) - the reference narrows its cells by the same kind of bound. */
    uint64_t const shorter_side = q_stats->longest < c_stats->longest ? q_stats->longest : c_stats->longest;
    d->narrow = d->objective == szs_objective_global_k              ? reach < 32000
                : d->objective == szs_objective_local_saturating_k ? (shorter_side + 3) * magnitude < 32000
                                                                    : 0;
    /* Two cells per VALU operation (hip/weighted_packed.hip) when EVERY DP value fits 16 bits, not just the parked ones:
     * the same two bounds cover all cells and tracks - the reach is a bound on any sum of (rows + columns + 3) costs. */
    if (d->maximise)
        for (int i = 0; i < 256; ++i) d->classes = engine->byte_to_class[i] >= d->classes ? (uint32_t)engine->byte_to_class[i] + 1 : d->classes;
    d->packed = d->maximise && d->narrow && d->classes <= 32 && szs_tuning_get(szs_knob_packed_k) != 0;
    d->packed_local = d->objective == szs_objective_local_saturating_k;
    /* The team tier (hip/weighted_teams.hip) scores what the packed kernel scores - and local alignments up to twice its
     * range - with a pair spread over the lanes of a DPP row; its cells are half-float patterns (three-input maxima) while
     * the bound allows, unsigned integers beyond (hip/team_core.hpp). */
    int team_capable = 0;
    if (d->maximise && !d->wide_cells && d->classes <= 32 && szs_tuning_get(szs_knob_packed_k) != 0 &&
        (d->objective == szs_objective_global_k || d->objective == szs_objective_local_saturating_k)) {
        uint64_t const bound = d->packed_local ? (shorter_side + 3) * magnitude : reach; /* what bounds every H and track value */
        d->team_objective = d->packed_local;
        d->team_wide = bound >= szs_hip_weighted_team_reach_limit(d->team_objective, 0);
        team_capable = bound < szs_hip_weighted_team_reach_limit(d->team_objective, d->team_wide);
    }
    /* A Levenshtein engine with non-unit costs over bytes: the same tier over the negated costs, keyed by the batch's own dense
     * alphabet (engine->uniform_classes, counted on the device for this very call - 0 when the inputs were not tapes), with
     * 16-bit cells up to a reach of 64000 where the 32-bit kernel of weighted.hip spends 3 / 7 VALU operations per cell
     * (the reference: 4 x u8 / 2 x u16 register kernels and u8 / u16 warp cells, cuda.cuh:2939-3128, tiers :1842-1854). */
    if (d->objective == szs_objective_distance_k && !d->use_myers && !d->wide_cells && engine->uniform_classes &&
        szs_tuning_get(szs_knob_packed_k) != 0) {
        d->team_objective = 2, d->classes = engine->uniform_classes;
        d->team_wide = reach >= szs_hip_weighted_team_reach_limit(2, 0);
        team_capable = reach < szs_hip_weighted_team_reach_limit(2, d->team_wide);
    }

    /* ---- orientation and tier.  Every kernel puts ONE side on workgroups / band chains (its "queries") and the other
     * on lanes / columns (its "candidates"); which real side plays which role is free - gap costs apply to both strings
     * alike and a swapped class table is its transpose - so a cycle model of both tiers (plan.c) is evaluated for both
     * orientations and the cheaper one runs.  1024 queries x 1 candidate thus become 1 workgroup row of 1024 lanes
     * instead of 1024 workgroups with one live lane each.  Symmetric calls have nothing to swap. */
    szs_plan_orient(d->use_myers ? 0xFFFFFFFFu : 0 /* bit-parallel at any length, bytes and codepoints alike */, d->use_myers && !runes, !engine->is_linear,
                    !d->maximise, team_capable, symmetric, q_stats, c_stats, szs_hip_systolic_band_rows(), &d->tier, &d->transposed);
    /* the `tier` knob: `systolic` on a unit-cost engine means the DP recurrences, `chain` the bit-parallel chain */
    if (d->tier == SZS_TIER_MYERS_CHAIN && szs_tuning_get(szs_knob_tier_k) == SZS_TIER_SYSTOLIC) d->tier = SZS_TIER_SYSTOLIC;
    if (force_lanes) d->tier = SZS_TIER_LANES;

    szs_side_stats_t const *kq = d->transposed ? c_stats : q_stats, *kc = d->transposed ? q_stats : c_stats;
    uint32_t const *kq_variants = d->transposed ? c_variants : q_variants;
    d->kq_count = kq->count, d->kc_count = kc->count;
    d->layout = (symmetric ? SZS_LAYOUT_SYMMETRIC : 0) | (d->transposed ? SZS_LAYOUT_TRANSPOSED : 0);
    d->plan.longest_query = kq->longest, d->plan.longest_candidate = kc->longest, d->plan.cells = cells;
    memcpy(d->variant_counts, kq_variants, sizeof(d->variant_counts));
    szs_plan_groups(kq_variants, &d->plan);

    if (d->wide_cells) /* one tier only: the anti-diagonal walker with 64-bit cells, whatever the shape of the batch */
        d->tier = SZS_TIER_LANES, d->packed = 0, d->narrow = 0;
    d->team = 0;
    if (team_capable && d->tier == SZS_TIER_LANES) d->team = team_shape_for(!engine->is_linear, d->classes, kq, kc);
    if (d->team) d->packed = 1, d->packed_local = d->team_objective == 1; /* the profile reports 16-bit cells either way */
    else d->team_wide = 0, d->team_objective = 0;

    /* The systolic tier scores every engine family with its weighted recurrences, so it needs the cost model and its
     * own workspace; a job with too many pairs in flight for that workspace stays on the lanes tier. */
    if (d->tier == SZS_TIER_SYSTOLIC &&
        (!szs_hip_systolic_workspace_bytes(!engine->is_linear, d->kq_count, d->kc_count, d->plan.longest_query, d->plan.longest_candidate,
                                           &d->systolic_control_bytes, &d->systolic_parked_bytes) ||
         d->systolic_control_bytes + d->systolic_parked_bytes > ((size_t)32 << 30)))
        d->tier = SZS_TIER_LANES;
    if (d->tier == SZS_TIER_MYERS_CHAIN &&
        (!szs_hip_myers_chain_workspace_bytes(d->kq_count, d->kc_count, d->plan.longest_query, d->plan.longest_candidate,
                                              &d->systolic_control_bytes, &d->systolic_parked_bytes) ||
         d->systolic_control_bytes + d->systolic_parked_bytes > ((size_t)32 << 30)))
        d->tier = SZS_TIER_LANES;
    d->kq_symbols = kq->symbols;
    (void)ranks;
    (void)error_message;
    d->valid = 1;
    return sz_success_k;
}

/**
 *  The bit-parallel width groups of a unit-cost call as ONE persistent launch (hip/myers_queue.hip) - each used to be a launch
 *  with a tail of its own.  Bytes always can; codepoints when the device renumbered the batch (`d->alphabet`) and the alphabet
 *  leaves every query a table.  `ranks`: the lengths at 33 ranks of the CALLER's sides (the device planner's summary), or NULL
 *  when the host planner has yet to sort (it plans the queue itself afterwards).  The `queue` knob: 0 never, 1 whenever possible.
 *
 *  Automatic: batches of SKEWED lengths - the longest query at least 2.5 times the mean, two width groups or more.  Those are
 *  the calls whose per-width launches end in tails (config 5, its shares on several GPUs, lines of text).  A batch of one
 *  length class that merely straddles two or three widths keeps its launches: its work items would all be alike, a handful
 *  per workgroup, and the last round of them runs the device half empty (1024 x 1024 x 500 bytes: 112 against 90 TCUPS).
 */
void szs_call_decide_queue(szs_engine_s const *engine, szs_decision_t *d, uint32_t const (*ranks)[SZS_PLAN_RANK_SAMPLES + 1]) {
    int const queue_knob = szs_tuning_get(szs_knob_queue_k);
    unsigned bit_parallel_groups = 0;
    for (unsigned g = 0; g < d->plan.groups_count; ++g) bit_parallel_groups += d->plan.groups[g].variant != 0;
    int const skewed = d->kq_count && (uint64_t)d->plan.longest_query * d->kq_count * 2u >= d->kq_symbols * 5u;
    d->use_queue = d->use_myers && (!d->runes || d->alphabet) && d->tier == SZS_TIER_LANES && !d->wide_cells && queue_knob != 0 &&
                   !engine->queue_refused && (queue_knob > 0 ? bit_parallel_groups >= 1u : bit_parallel_groups >= 2u && skewed);
    if (!d->use_queue || !ranks) return;
    d->plan.has_ranks = 1;
    memcpy(d->plan.rank_lengths[0], ranks[d->transposed ? 1 : 0], sizeof(d->plan.rank_lengths[0]));
    memcpy(d->plan.rank_lengths[1], ranks[d->transposed ? 0 : 1], sizeof(d->plan.rank_lengths[1]));
    szs_plan_queue(&d->plan, d->kq_count, d->kc_count, d->runes ? d->alphabet : 0u, szs_hip_levenshtein_myers_queue_table_bytes(d->runes), &d->queue);
    if (!d->queue.items_total) d->use_queue = 0; /* an alphabet too rich for the tables: the per-width launches */
}

/** Does the lanes tier of this decision run a kernel that reads the cost model / the weighted strip workspace? */
int szs_decision_has_variant_zero(szs_decision_t const *d) {
    for (unsigned g = 0; g < d->plan.groups_count; ++g)
        if (d->plan.groups[g].variant == 0) return 1;
    return 0;
}

/** Calls of ONE width group are speculated (launched behind the planner on the previous call's shape, §3).  Calls of the one
 *  persistent launch of hip/myers_queue.hip are NOT, although its queue addresses the sorted refs by position and would be valid
 *  for any batch of the same counts per width: measured (round 4), the host's round trip it saves (16 us) came back as a longer
 *  launch (an eighth of config 5: 1.610 ms planned, 1.612 speculated; codepoints 1.061 / 1.049) - the planner's own 35 - 45 us for
 *  3,500 strings are what such a call waits for, not the host.  (Several launches released by one event reach the device in no
 *  particular order: those calls are planned and waited for as well.) */
int szs_decision_is_one_launch(szs_decision_t const *d) { return d->plan.groups_count == 1; }

static sz_status_t upload_model(szs_engine_s *engine, szs_decision_t const *d, int device, hipStream_t stream,
                                char const **error_message) {
    int const per_call = d->team && d->team_objective == 2; /* the byte classes are the batch's: a new model every call */
    if (!per_call && engine->model_uploaded_device == device && engine->model_uploaded_transposed == d->transposed) return sz_success_k;
    sz_status_t const status = szs_buffer_reserve(&engine->device_model, szs_memory_device_k, device, sizeof(szs_cost_model_t), error_message);
    if (status != sz_success_k) return status;
    fill_cost_model(engine, d->transposed, &engine->host_model); /* lives in the engine: the copy may complete later */
    hipError_t const error = hipMemcpyAsync(engine->device_model.pointer, &engine->host_model, sizeof(szs_cost_model_t), hipMemcpyHostToDevice, stream);
    if (error != hipSuccess) return szs_report_hip(error, error_message);
    engine->model_uploaded_device = per_call ? -1 : device, engine->model_uploaded_transposed = d->transposed;
    return sz_success_k;
}

/** Size of the weighted lanes kernels' strip workspace for this decision. */
static size_t weighted_boundary_bytes(szs_engine_s const *engine, szs_decision_t const *d) {
    if (d->team)
        return szs_hip_weighted_team_workspace_bytes(d->team_objective, !engine->is_linear, d->team_wide, d->team, d->classes, d->kq_count, d->kc_count,
                                                     d->plan.longest_candidate);
    return d->packed ? szs_hip_weighted_packed_boundary_bytes(d->packed_local, !engine->is_linear, d->classes, d->kq_count, d->kc_count,
                                                              d->plan.longest_candidate)
                     : szs_hip_weighted_boundary_bytes(d->objective, !engine->is_linear, d->narrow, d->kq_count, d->kc_count,
                                                       d->plan.longest_candidate);
}

/**
 *  Workspaces and the cost model, sized for the kernels that CAN launch - and only those: the strip kernel of long byte
 *  queries parks 0.25 B per column and lane where the weighted kernels park 4, so a unit-cost call over 100 KB strings
 *  reserves ~5 GB, not ~77.
 */
sz_status_t szs_call_prepare(szs_engine_s *engine, szs_decision_t const *d, int device, hipStream_t stream, char const **error_message) {
    sz_status_t status = sz_success_k;
    hipError_t error = hipSuccess;
    int const chained = d->tier == SZS_TIER_SYSTOLIC || d->tier == SZS_TIER_MYERS_CHAIN;
    if (chained) {
        /* The control block is zeroed when it is (re)allocated and never again: its words carry the epoch of the launch
         * that wrote them, so a launch neither needs nor waits for a fill (hip/kernels.h). */
        int const fresh = !engine->device_systolic.pointer || engine->device_systolic.capacity < d->systolic_control_bytes ||
                          engine->device_systolic.device != device; /* the reserve below will (re)allocate */
        status = szs_buffer_reserve(&engine->device_systolic, szs_memory_device_k, device, d->systolic_control_bytes, error_message);
        if (status != sz_success_k) return status;
        if (fresh || engine->systolic_epoch >= 0xFFFFFFF0u) {
            error = hipMemsetAsync(engine->device_systolic.pointer, 0, engine->device_systolic.capacity, stream);
            if (error != hipSuccess) return szs_report_hip(error, error_message);
            engine->systolic_epoch = 0;
        }
        engine->systolic_epoch++;
        status = szs_buffer_reserve(&engine->device_boundary, szs_memory_device_k, device, d->systolic_parked_bytes, error_message);
        if (status != sz_success_k) return status;
        if (d->tier == SZS_TIER_SYSTOLIC) return upload_model(engine, d, device, stream, error_message);
        return sz_success_k;
    }
    if (d->wide_cells) {
        size_t const bytes = szs_hip_wide_workspace_bytes(!engine->is_linear, d->kq_count, d->kc_count, d->plan.longest_query,
                                                          d->plan.longest_candidate);
        status = szs_buffer_reserve(&engine->device_boundary, szs_memory_device_k, device, bytes, error_message);
        if (status != sz_success_k) return status;
        return upload_model(engine, d, device, stream, error_message);
    }
    if (d->use_queue) { /* the ticket counter of the persistent launch: zeroed when allocated, then only ever counted up */
        status = szs_buffer_reserve(&engine->device_queue, szs_memory_device_k, device, 256, error_message);
        if (status != sz_success_k) return status;
        if (engine->queue_zeroed != engine->device_queue.pointer) {
            error = hipMemsetAsync(engine->device_queue.pointer, 0, 256, stream);
            if (error != hipSuccess) return szs_report_hip(error, error_message);
            engine->queue_zeroed = engine->device_queue.pointer, engine->queue_tickets = 0;
        }
    }
    int const variant_zero = szs_decision_has_variant_zero(d);
    size_t boundary_bytes = 0;
    if (d->use_myers && d->runes && variant_zero) /* codepoint queries beyond 2048 runes: the strip kernel's parked deltas */
        boundary_bytes = szs_hip_levenshtein_myers_banded_runes_bytes(d->kq_count, d->kc_count, d->plan.longest_candidate);
    if (!d->use_myers || (d->runes && variant_zero && !boundary_bytes)) { /* the weighted kernels: every non-unit engine (and
                                                                             long rune queries on a device without the LDS) */
        status = upload_model(engine, d, device, stream, error_message);
        if (status != sz_success_k) return status;
        boundary_bytes = weighted_boundary_bytes(engine, d);
    }
    else if (d->banded && variant_zero) /* byte queries beyond 64 words: the strip kernel's parked deltas, nothing else */
        boundary_bytes = szs_hip_levenshtein_myers_banded_bytes(d->kq_count, d->kc_count, d->plan.longest_candidate);
    if (boundary_bytes) status = szs_buffer_reserve(&engine->device_boundary, szs_memory_device_k, device, boundary_bytes, error_message);
    return status;
}

/**
 *  Auxiliary streams, per device, for the whole process: creating a HIP stream costs 7-12 ms on this runtime (a hardware queue
 *  each, up to GPU_MAX_HW_QUEUES) - seven of them were 85 ms on the first mixed-length call of EVERY engine when engines owned
 *  theirs.  They are created on demand, as many as a call needs, and live as long as the library.  Engines that share them
 *  still order their own work with their own events; concurrent calls of different engines merely take turns on a stream.
 */
#define SZS_AUX_DEVICES 64
static struct {
    pthread_mutex_t lock;
    hipStream_t streams[SZS_AUX_DEVICES][SZS_AUX_STREAMS];
    unsigned created[SZS_AUX_DEVICES];
} aux_pool = {PTHREAD_MUTEX_INITIALIZER, {{0}}, {0}};

static hipError_t szs_aux_streams(int device, unsigned wanted, hipStream_t *streams) {
    if (device < 0 || device >= SZS_AUX_DEVICES || wanted > SZS_AUX_STREAMS) return hipErrorInvalidValue;
    hipError_t error = hipSuccess;
    pthread_mutex_lock(&aux_pool.lock);
    while (aux_pool.created[device] < wanted && error == hipSuccess) { /* the caller has made `device` current */
        error = hipStreamCreateWithFlags(&aux_pool.streams[device][aux_pool.created[device]], hipStreamNonBlocking);
        if (error == hipSuccess) ++aux_pool.created[device];
    }
    for (unsigned i = 0; i < wanted && error == hipSuccess; ++i) streams[i] = aux_pool.streams[device][i];
    pthread_mutex_unlock(&aux_pool.lock);
    return error;
}

/** Launches of one decision over device refs in kernel roles.  Returns the first launch error; counts launches. */
hipError_t szs_call_enqueue(szs_engine_s *engine, szs_decision_t const *d, int device, szs_string_ref_t const *query_refs,
                          szs_string_ref_t const *candidate_refs, void *device_results, size_t device_stride, hipStream_t stream,
                          szs_ref_guard_t const *guard /* refs of an earlier call: validate in the kernels; else NULL */,
                          uint32_t *launches, uint32_t *cell_bits, sz_status_t *status, char const **error_message) {
    int launch_error = 0;
    szs_cost_model_t const *const model = (szs_cost_model_t const *)engine->device_model.pointer;
    *status = sz_success_k;
    if (d->tier == SZS_TIER_SYSTOLIC) { /* one launch for the whole cross-product, whatever the planner's groups */
        launch_error = szs_hip_systolic_scores(d->objective, !engine->is_linear, model, query_refs, d->kq_count, candidate_refs,
                                               d->kc_count, d->plan.longest_query, d->plan.longest_candidate, (int64_t *)device_results,
                                               device_stride, d->layout, engine->device_systolic.pointer,
                                               engine->device_boundary.pointer, engine->systolic_epoch, stream);
        ++*launches, *cell_bits = 32;
        return (hipError_t)launch_error;
    }
    if (d->tier == SZS_TIER_MYERS_CHAIN) {
        launch_error = szs_hip_myers_chain(query_refs, d->kq_count, candidate_refs, d->kc_count, d->plan.longest_query,
                                           d->plan.longest_candidate, (uint64_t *)device_results, device_stride, d->layout,
                                           engine->device_systolic.pointer, engine->device_boundary.pointer, engine->systolic_epoch, stream);
        ++*launches;
        return (hipError_t)launch_error;
    }
    if (d->wide_cells) {
        launch_error = szs_hip_wide_scores(d->objective, !engine->is_linear, model, query_refs, d->kq_count, candidate_refs, d->kc_count,
                                           d->plan.longest_query, d->plan.longest_candidate, (int64_t *)device_results, device_stride,
                                           d->layout, engine->device_boundary.pointer, stream);
        ++*launches, *cell_bits = 64;
        return (hipError_t)launch_error;
    }
    /* Persistent kernels address their work items with 32 bits: cross-products beyond that are cut along the query axis. */
    uint64_t candidate_blocks = ((uint64_t)d->kc_count + SZS_CANDIDATES_PER_WORKGROUP - 1) / SZS_CANDIDATES_PER_WORKGROUP;
    uint64_t queries_most = 0xFFFFFFF0ull / (candidate_blocks ? candidate_blocks : 1);
    if (d->team) { /* the team tier's item is a PAIR of queries x the 256 / lanes candidates of a workgroup (weighted_teams.hip) */
        uint64_t per_block = szs_hip_weighted_team_candidates_per_item(d->team); /* 256 / lanes; 512 / lanes for the wave-wide teams */
        if (!per_block) per_block = 256u / (d->team / 10000u ? d->team / 10000u : 1u);
        candidate_blocks = ((uint64_t)d->kc_count + per_block - 1) / per_block;
        queries_most = 2 * (0xFFFFFFF0ull / (candidate_blocks ? candidate_blocks : 1)); /* even: the pairs of a cut stay pairs */
    }
    uint32_t const queries_per_launch = queries_most < 0xFFFFFFF0ull ? (uint32_t)queries_most : 0xFFFFFFF0u;

    /* A batch of mixed lengths is several launches (one per bit-vector width), and every launch ends in a tail: its last,
     * longest pairs hold a few wavefronts while the rest of the chip idles - on config 5 the two widest launches kept 0.7
     * of a wavefront per SIMD resident (profiles/r02/pmc_configs.json).  The launches are independent (disjoint result
     * cells), so they fan out over the engine's auxiliary streams and fill each other's tails; only the launches that
     * share the strip workspace (`device_boundary`) stay on the scope's stream, in order. */
    /* Unit-cost byte calls of several widths: ONE persistent launch scores every bit-parallel group (hip/myers_queue.hip); what
     * is left for the loop below is the strip kernel's group, if any.  (Not behind a guard: re-used plans are one short launch.) */
    int const queued = d->use_queue && !guard && d->queue.items_total != 0;
    int const fan_out = d->plan.groups_count > 1 && szs_tuning_get(szs_knob_streams_k) != 0 && !queued;
    /* no more streams than the process has hardware queues (tuning.c): streams that share a queue run in the queue's order */
    unsigned const queues = (unsigned)szs_tuning_get(szs_knob_queues_k);
    unsigned const most_aux = queues ? (queues - 1 < SZS_AUX_STREAMS ? queues - 1 : (unsigned)SZS_AUX_STREAMS) : 0u;
    unsigned const aux_used = !fan_out ? 0u : d->plan.groups_count - 1 < most_aux ? d->plan.groups_count - 1 : most_aux;
    engine->last_streams = aux_used + 1;
    hipError_t error = hipSuccess;
    if (fan_out) {
        if (engine->aux_device != device) {
            if (engine->aux_device >= 0) return hipErrorInvalidDevice; /* release_device_state() precedes a change of device */
            for (int i = 0; i < SZS_AUX_STREAMS && error == hipSuccess; ++i) error = hipEventCreateWithFlags(&engine->aux_done[i], hipEventDisableTiming);
            if (error == hipSuccess) error = hipEventCreateWithFlags(&engine->fork_event, hipEventDisableTiming);
            if (error != hipSuccess) return error;
            engine->aux_device = device;
        }
        error = szs_aux_streams(device, aux_used, engine->aux_streams);
        if (error != hipSuccess) return error;
        error = hipEventRecord(engine->fork_event, stream);
        for (unsigned i = 0; i < aux_used && error == hipSuccess; ++i) error = hipStreamWaitEvent(engine->aux_streams[i], engine->fork_event, 0);
        if (error != hipSuccess) return error; /* nothing has been launched on the auxiliary streams */
    }
    /* Every width group's launch shape, and the order the launches go out in: LONGEST PAIR FIRST.  What a launch cannot go
     * under is its longest pair - columns x words PER LANE, one dependent instruction after the other - and the first launch
     * submitted takes every free wavefront slot: the widest group (64 words over eight lanes: many workgroups, short pairs)
     * used to go first and the 20-word launch - one lane per pair, the longest pairs of all - got its first wavefront 0.4 ms
     * into a 2 ms call.  Groups that need the workspace keep their place at the front, in order, on the scope's stream; the
     * short launch (thousands of workgroups that live microseconds) goes last and fills what the others leave. */
    szs_launch_shape_t shapes[SZS_PLAN_MAX_GROUPS];
    unsigned order[SZS_PLAN_MAX_GROUPS];
    szs_plan_launch_order(&d->plan, d->use_myers, d->runes, candidate_blocks, szs_tuning_get(szs_knob_split_k), shapes, order);
    /* Launches that need no workspace are dealt over {scope's stream, auxiliary streams} back and forth (0 .. n, n .. 0, ...):
     * the second launch of a stream is the lightest one left. */
    unsigned next_lane = 0;
    for (unsigned turn_of = 0; turn_of < d->plan.groups_count && !launch_error && *status == sz_success_k; ++turn_of) {
        unsigned const g = order[turn_of];
        szs_plan_group_t const *group = &d->plan.groups[g];
        szs_launch_shape_t const shape = shapes[g];
        int const uses_workspace = group->variant == 0;
        if (queued && group->variant) continue; /* in the queue */
        unsigned lane = 0;
        if (fan_out && !uses_workspace) {
            unsigned const turn = next_lane / (aux_used + 1), place = next_lane % (aux_used + 1);
            lane = turn & 1 ? aux_used - place : place, ++next_lane;
        }
        hipStream_t const target = lane ? engine->aux_streams[lane - 1] : stream;
        for (uint32_t done = 0; done < group->count && !launch_error && *status == sz_success_k; done += queries_per_launch) {
            szs_string_ref_t const *const queries = query_refs + group->first + done;
            uint32_t const count = group->count - done < queries_per_launch ? group->count - done : queries_per_launch;
            if (group->variant && d->runes) {
                launch_error = group->variant == SZS_MYERS_SHORT_WORDS
                                   ? szs_hip_levenshtein_myers_runes(queries, count, candidate_refs, d->kc_count, (uint64_t *)device_results,
                                                                     device_stride, d->layout, d->alphabet, target)
                               : shape.lanes ? szs_hip_levenshtein_myers_runes_split(shape.words, shape.lanes, queries, count, candidate_refs, d->kc_count,
                                                                               (uint64_t *)device_results, device_stride, d->layout, d->alphabet,
                                                                               (uint64_t)group->count * candidate_blocks < 1024 ? 64u : 256u /* measured: an eighth of config
                                                                               5u 2.01 -> 1.72 ms, a quarter 2.77 -> 2.36, the whole unchanged; under
                                                                               4096 the whole of it loses 12 % */, target)
                                       : szs_hip_levenshtein_myers_runes_long(group->variant, queries, count, candidate_refs, d->kc_count,
                                                                              (uint64_t *)device_results, device_stride, d->layout, d->alphabet, target);
                if (launch_error == (int)hipErrorNotSupported) { /* no LDS for the rune table: the rune-keyed DP kernel, whose
                                                                    workspace is reserved here, on the one path that needs it -
                                                                    and on the scope's stream, like every user of that workspace */
                    launch_error = 0;
                    *status = upload_model(engine, d, device, stream, error_message);
                    if (*status == sz_success_k)
                        *status = szs_buffer_reserve(&engine->device_boundary, szs_memory_device_k, device,
                                                     szs_hip_weighted_boundary_bytes(d->objective, !engine->is_linear, 0, d->kq_count, d->kc_count,
                                                                                     d->plan.longest_candidate), error_message);
                    if (*status != sz_success_k) break;
                    *cell_bits = 32;
                    launch_error = szs_hip_weighted_scores(d->objective, !engine->is_linear, 0, (szs_cost_model_t const *)engine->device_model.pointer,
                                                           queries, count, candidate_refs, d->kc_count, d->plan.longest_candidate,
                                                           (int64_t *)device_results, device_stride, d->layout, engine->device_boundary.pointer, stream);
                }
            }
            else if (group->variant && shape.lanes)
                launch_error = szs_hip_levenshtein_myers_split(shape.words, shape.lanes, queries, count, candidate_refs, d->kc_count,
                                                               (uint64_t *)device_results, device_stride, d->layout, guard, target);
            else if (group->variant)
                launch_error = szs_hip_levenshtein_myers(group->variant, queries, count, candidate_refs, d->kc_count, (uint64_t *)device_results,
                                                         device_stride, d->layout, guard, target);
            else if (d->banded)
                launch_error = szs_hip_levenshtein_myers_banded(queries, count, candidate_refs, d->kc_count, d->plan.longest_candidate,
                                                                (uint64_t *)device_results, device_stride, d->layout,
                                                                engine->device_boundary.pointer, target);
            else if (d->use_myers && d->runes) { /* codepoints beyond 2048 runes: strips with a rune table per strip */
                launch_error = szs_hip_levenshtein_myers_banded_runes(queries, count, candidate_refs, d->kc_count, d->plan.longest_candidate,
                                                                      (uint64_t *)device_results, device_stride, d->layout,
                                                                      engine->device_boundary.pointer, target);
                if (launch_error == (int)hipErrorNotSupported) { /* as above: the rune-keyed DP kernel and its workspace */
                    launch_error = 0;
                    *status = upload_model(engine, d, device, stream, error_message);
                    if (*status == sz_success_k)
                        *status = szs_buffer_reserve(&engine->device_boundary, szs_memory_device_k, device,
                                                     szs_hip_weighted_boundary_bytes(d->objective, !engine->is_linear, 0, d->kq_count, d->kc_count,
                                                                                     d->plan.longest_candidate), error_message);
                    if (*status != sz_success_k) break;
                    *cell_bits = 32;
                    launch_error = szs_hip_weighted_scores(d->objective, !engine->is_linear, 0, (szs_cost_model_t const *)engine->device_model.pointer,
                                                           queries, count, candidate_refs, d->kc_count, d->plan.longest_candidate,
                                                           (int64_t *)device_results, device_stride, d->layout, engine->device_boundary.pointer, stream);
                }
            }
            else if (d->team) {
                *cell_bits = 16;
                launch_error = szs_hip_weighted_team_scores(d->team_objective, !engine->is_linear, d->team_wide, d->team, d->classes, model, queries, count,
                                                            candidate_refs, d->kc_count, d->plan.longest_candidate, (int64_t *)device_results,
                                                            device_stride, d->layout, engine->device_boundary.pointer, target);
            }
            else if (d->packed) {
                *cell_bits = 16;
                launch_error = szs_hip_weighted_packed_scores(d->packed_local, !engine->is_linear, d->classes, model, queries, count, candidate_refs,
                                                              d->kc_count, d->plan.longest_candidate, (int64_t *)device_results, device_stride,
                                                              d->layout, engine->device_boundary.pointer, target);
            }
            else {
                *cell_bits = 32;
                launch_error = szs_hip_weighted_scores(d->objective, !engine->is_linear, d->narrow, model, queries, count, candidate_refs,
                                                       d->kc_count, d->plan.longest_candidate, (int64_t *)device_results, device_stride,
                                                       d->layout, engine->device_boundary.pointer, target);
            }
            if (!launch_error) ++*launches;
        }
    }
    if (queued && !launch_error && *status == sz_success_k) {
        uint32_t taken = 0;
        uint64_t *trace = NULL; /* `trace` knob: where every workgroup's begin / end ticks go (szs_call_finish() prints their spread) */
        if (szs_tuning_get(szs_knob_trace_k) > 0 &&
            szs_buffer_reserve(&engine->device_queue_trace, szs_memory_device_k, device,
                               7 * sizeof(uint64_t) * (size_t)szs_hip_levenshtein_myers_queue_grid(d->queue.items_total, d->runes), NULL) == sz_success_k)
            trace = (uint64_t *)engine->device_queue_trace.pointer;
        /* a query that fits no table of the kernel (the plan's job to prevent) raises this flag in pinned memory: szs_engine_cross
         * looks after the call's wait and scores the batch with the per-width launches instead */
        uint32_t *const unfit = (uint32_t *)((char *)engine->pinned_summary.pointer + 960);
        engine->queue_unfit_sequence = ++engine->plan_sequence;
        launch_error = szs_hip_levenshtein_myers_queue(&d->queue, query_refs, candidate_refs, (uint64_t *)device_results, device_stride, d->layout,
                                                       (uint32_t *)engine->device_queue.pointer, engine->queue_tickets, &taken, trace,
                                                       d->runes ? d->alphabet : 0u, unfit, engine->queue_unfit_sequence, stream);
        engine->queue_tickets += taken; /* wraps with the counter */
        if (!launch_error) ++*launches, engine->last_queued = 1; /* (a guarded re-use of the plan takes the per-width launches: not this) */
    }
    if (fan_out) /* join, also after a failed launch: whatever was enqueued anywhere is drained by the wait on the scope's stream */
        for (unsigned i = 0; i < aux_used; ++i) {
            hipError_t joined = hipEventRecord(engine->aux_done[i], engine->aux_streams[i]);
            if (joined == hipSuccess) joined = hipStreamWaitEvent(stream, engine->aux_done[i], 0);
            if (joined != hipSuccess) { /* cannot order the streams: wait here, so that nothing outlives the call */
                (void)hipStreamSynchronize(engine->aux_streams[i]);
                if (!launch_error) launch_error = (int)joined;
            }
        }
    return (hipError_t)launch_error;
}

/* ---- one call ---------------------------------------------------------------------------------------------------------- */

/* (szs_call_t: dispatch_internal.h) */

/* ---- roctx ranges (`roctx` knob): the host phases of a call as nested ranges a `rocprofv3 --marker-trace` timeline shows
 *      next to the kernels - the counterpart of the reference's NVTX-free but timer-instrumented executors.
 *      The marker library is looked up at run time: the product library does not link it, and without the knob nothing is
 *      loaded and a call pays one relaxed load. */
static struct {
    pthread_once_t once;
    int (*push)(char const *);
    int (*pop)(void);
} roctx = {PTHREAD_ONCE_INIT, NULL, NULL};

static void roctx_resolve(void) {
    char const *const names[] = {"librocprofiler-sdk-roctx.so.1", "librocprofiler-sdk-roctx.so", "libroctx64.so.4", "libroctx64.so"};
    for (unsigned i = 0; i < sizeof(names) / sizeof(names[0]) && !roctx.push; ++i) {
        void *const library = dlopen(names[i], RTLD_NOW | RTLD_GLOBAL);
        if (!library) continue;
        roctx.push = (int (*)(char const *))dlsym(library, "roctxRangePushA");
        roctx.pop = (int (*)(void))dlsym(library, "roctxRangePop");
        if (!roctx.push || !roctx.pop) roctx.push = NULL, roctx.pop = NULL;
    }
}

static char const *const phase_names[6] = {"szs: inputs and buffers", "szs: plan", "szs: decide and prepare", "szs: enqueue",
                                           "szs: wait for the device", "szs: profile"};

static void ranges_begin(szs_call_t *call) {
    call->ranges = 0;
    if (szs_tuning_get(szs_knob_roctx_k) <= 0) return;
    (void)pthread_once(&roctx.once, roctx_resolve);
    if (!roctx.push) return;
    (void)roctx.push("szs_engine_cross"), (void)roctx.push(phase_names[0]);
    call->ranges = 2;
}

static void ranges_end(szs_call_t *call) {
    for (; call->ranges > 0; --call->ranges) (void)roctx.pop();
}

/** The END of phase `index`: accounts its time (`trace` knob) and opens the next phase's range (`roctx` knob). */
void szs_call_phase(szs_call_t *call, int index) {
    if (call->ranges == 2) {
        (void)roctx.pop();
        if (index + 1 < 6) (void)roctx.push(phase_names[index + 1]);
        else call->ranges = 1;
    }
    if (!call->trace) return;
    double const now = now_milliseconds();
    call->phases[index] += now - call->phase_started, call->phase_started = now;
}

/** Everything after the last launch: stop event, stall flag, copy-out, THE wait, profile.  `enqueued` failures and every
 *  failure in here drain the stream before the status leaves the library. */
sz_status_t szs_call_finish(szs_call_t *call, szs_decision_t const *d, hipError_t error, sz_status_t status, uint32_t launches,
                          uint32_t cell_bits, uint64_t query_symbols, uint64_t candidate_symbols, int *stalled) {
    szs_engine_s *engine = call->engine;
    hipStream_t const stream = call->stream;
    int const chained = d->tier == SZS_TIER_SYSTOLIC || d->tier == SZS_TIER_MYERS_CHAIN;
    uint64_t *const stall_flag = (uint64_t *)engine->pinned_summary.pointer + 64; /* behind the planner's summary */
    *stall_flag = 0;
    if (error == hipSuccess && status == sz_success_k) error = hipEventRecord(engine->event_stop, stream);
    if (error == hipSuccess && status == sz_success_k && chained)
        error = hipMemcpyAsync(stall_flag, (char *)engine->device_systolic.pointer + 8, sizeof(uint64_t), hipMemcpyDeviceToHost, stream);
    if (error == hipSuccess && status == sz_success_k && !call->direct) /* one strided copy back into the caller's host matrix
                                                                           (reference: cuMemcpy2DAsync, cuda.cuh:2205-2215) */
        error = hipMemcpy2DAsync(call->results, call->results_row_stride * sizeof(uint64_t), call->device_results,
                                 call->device_stride * sizeof(uint64_t), (size_t)call->c_count * sizeof(uint64_t), call->q_count,
                                 hipMemcpyDefault, stream);
    szs_call_phase(call, 3); /* launches enqueued */
    hipError_t const drained = hipStreamSynchronize(stream); /* the call is synchronous, like the reference's - also when it fails */
    /* (Round 6, measured and not kept: polling the stop event from the calling thread before blocking - config 2's call 193.9 us
     * either way, host overhead 14.0 against 13.6: the runtime's own wait already spins.  profiles/r06/wait_polling.txt) */
    szs_call_phase(call, 4); /* waiting for the device */
    if (status != sz_success_k || error != hipSuccess || drained != hipSuccess)
        engine->queue_zeroed = NULL, /* the host's mirror of the queue kernel's ticket counter may no longer match the device's (a launch
                                        counted but never run, or run but reported failed): szs_call_prepare() zeroes both before the next one */
            engine->fused_zeroed = NULL; /* ... and so may the planner's verdict counter (its parity elects the side that folds the two
                                            verdicts) and the `ready` words: szs_call_reserve_device_words() zeroes the 256 bytes again */
    if (status != sz_success_k) return status;
    if (error == hipSuccess) error = drained;
    if (error != hipSuccess) return szs_report_hip(error, call->error_message);
    *stalled = chained && *stall_flag == (((uint64_t)engine->systolic_epoch << 32) | 1);
    if (*stalled) return sz_success_k; /* the caller re-runs the batch on the lanes tier */

    float kernel_ms = 0;
    (void)hipEventElapsedTime(&kernel_ms, engine->event_start, engine->event_stop);
    szs_rocm_call_profile_t *profile = &engine->last_profile;
    uint64_t const q_count = call->q_count, c_count = call->c_count;
    uint64_t const pairs = call->symmetric ? q_count * (q_count + 1) / 2 : q_count * c_count;
    profile->kernel_milliseconds = kernel_ms;
    profile->cells = d->plan.cells;
    profile->pairs = pairs;
    /* Canonical pair-streaming bytes: len(q) + len(c) + two 4-byte offsets + one 8-byte result per pair; over the lower
     * triangle string i meets i + 1 partners as a query and n - i as a candidate: n + 1 times in all. */
    profile->algorithmic_bytes = call->symmetric ? (q_count + 1) * query_symbols : c_count * query_symbols + q_count * candidate_symbols;
    profile->algorithmic_bytes += pairs * 16;
    profile->unique_bytes = query_symbols + (call->symmetric ? 0 : candidate_symbols) +
                            (q_count + 1 + (call->symmetric ? 0 : c_count + 1)) * 4 + q_count * c_count * 8;
    profile->launches = launches;
    profile->tier = (uint32_t)d->tier;
    profile->transposed = (uint32_t)d->transposed;
    profile->cell_bits = cell_bits;
    profile->team = cell_bits == 16 ? d->team : 0;
    profile->team_wide = profile->team ? (uint32_t)d->team_wide : 0;
    profile->streams = engine->last_streams ? engine->last_streams : 1;
    profile->queue_items = engine->last_queued ? d->queue.items_total : 0, profile->queue_tiles = profile->queue_items ? d->queue.tiles_count : 0;
    engine->last_queued = 0;
    profile->longest_query = d->longest[0], profile->longest_candidate = d->longest[1];
    profile->host_milliseconds = now_milliseconds() - call->started;
    szs_call_phase(call, 5);
#ifdef SZS_PLAN_TIMESTAMPS
    if (call->trace) {
        unsigned long long const *stamps = (unsigned long long const *)engine->pinned_summary.pointer + 56;
        fprintf(stderr, "planner phases (10 ns ticks):");
        for (int k = 1; k < 8; ++k) fprintf(stderr, " %lld", (long long)(stamps[k] - stamps[k - 1]));
        fprintf(stderr, "\n");
    }
#endif
    if (call->trace && profile->queue_items && engine->device_queue_trace.pointer) { /* the persistent launch, workgroup by workgroup */
        unsigned const grid = szs_hip_levenshtein_myers_queue_grid(profile->queue_items, d->runes);
        uint64_t *const ticks = (uint64_t *)malloc(7 * sizeof(uint64_t) * (size_t)grid);
        if (ticks && hipMemcpy(ticks, engine->device_queue_trace.pointer, 7 * sizeof(uint64_t) * (size_t)grid, hipMemcpyDeviceToHost) == hipSuccess) {
            uint64_t first = ~0ull, last = 0, busy = 0, latest_begin = 0;
            for (unsigned w = 0; w < grid; ++w) {
                first = ticks[7 * w] < first ? ticks[7 * w] : first, last = ticks[7 * w + 1] > last ? ticks[7 * w + 1] : last;
                latest_begin = ticks[7 * w] > latest_begin ? ticks[7 * w] : latest_begin, busy += ticks[7 * w + 1] - ticks[7 * w];
            }
            unsigned done_by[10] = {0}; /* workgroups that had ended by k / 10 of the launch */
            for (unsigned w = 0; w < grid; ++w)
                for (unsigned k = 0; k < 10; ++k) done_by[k] += (ticks[7 * w + 1] - first) * 10 <= (uint64_t)(k + 1) * (last - first);
            fprintf(stderr, "queue launch: %u workgroups, %.1f us from the first begin to the last end (last begin at %.1f us), mean busy %.3f | ended by tenth:",
                    grid, (last - first) * 1e-2, (latest_begin - first) * 1e-2, (double)busy / ((double)(last - first) * grid));
            for (unsigned k = 0; k < 10; ++k) fprintf(stderr, " %u", done_by[k]);
            fprintf(stderr, "\n");
            for (unsigned shown = 0, w = 0; w < grid && shown < 12; ++w) /* the latest few, one by one */
                if ((last - ticks[7 * w + 1]) * 50 <= (last - first))
                    fprintf(stderr, "  latest: workgroup %u took %u items; its last was ticket %u, begun at %.1f us, ended at %.1f us; its longest was ticket %u: %.1f us\n", w,
                            (unsigned)ticks[7 * w + 2], (unsigned)ticks[7 * w + 3], (ticks[7 * w + 4] - first) * 1e-2, (ticks[7 * w + 1] - first) * 1e-2,
                            (unsigned)ticks[7 * w + 5], ticks[7 * w + 6] * 1e-2), ++shown;
            { /* the longest items of the launch, by tile */
                double longest_of_tile[SZS_QUEUE_MOST_TILES] = {0};
                for (unsigned w = 0; w < grid; ++w) {
                    if (!ticks[7 * w + 2]) continue;
                    unsigned tile = 0;
                    while (tile + 1 < d->queue.tiles_count && ticks[7 * w + 5] >= d->queue.tiles[tile + 1].first_item) ++tile;
                    if (ticks[7 * w + 6] * 1e-2 > longest_of_tile[tile]) longest_of_tile[tile] = ticks[7 * w + 6] * 1e-2;
                }
                for (unsigned tile = 0; tile < d->queue.tiles_count && tile < 24; ++tile)
                    if (longest_of_tile[tile] > 0) {
                        szs_queue_tile_t const *t = &d->queue.tiles[tile];
                        fprintf(stderr, "  tile %2u (items %u.., queries %u+%u, candidates %u..%u, %u per item x %u queries, %u lanes x %u words): longest item seen %.1f us\n", tile,
                                t->first_item, t->query_first, t->query_count, t->candidate_first, t->candidate_end, t->candidates_per_item, t->queries_per_item, t->lanes,
                                t->words_per_lane, longest_of_tile[tile]);
                    }
            }
            /* what the LAST workgroups were holding: the tiles of the items that ended in the last twentieth of the launch */
            unsigned late_of_tile[SZS_QUEUE_MOST_TILES] = {0};
            double late_ms_of_tile[SZS_QUEUE_MOST_TILES] = {0};
            for (unsigned w = 0; w < grid; ++w) {
                if ((last - ticks[7 * w + 1]) * 20 > (last - first) || !ticks[7 * w + 2]) continue;
                unsigned tile = 0;
                while (tile + 1 < d->queue.tiles_count && ticks[7 * w + 3] >= d->queue.tiles[tile + 1].first_item) ++tile;
                late_of_tile[tile]++, late_ms_of_tile[tile] += (ticks[7 * w + 1] - ticks[7 * w + 4]) * 1e-5;
            }
            for (unsigned tile = 0; tile < d->queue.tiles_count; ++tile)
                if (late_of_tile[tile]) {
                    szs_queue_tile_t const *t = &d->queue.tiles[tile];
                    fprintf(stderr, "  late: %3u workgroups ended on tile %2u (items %u..; queries %u+%u, candidates %u..%u, %u per item x %u queries, %u lanes x %u words%s): their last item took %.3f ms on average, began at %.0f %% of the queue\n",
                            late_of_tile[tile], tile, t->first_item, t->query_first, t->query_count, t->candidate_first, t->candidate_end, t->candidates_per_item,
                            t->queries_per_item, t->lanes, t->words_per_lane, t->flags & SZS_QUEUE_TILE_SPARSE ? ", sparse" : "",
                            late_ms_of_tile[tile] / late_of_tile[tile], 100.0 * t->first_item / d->queue.items_total);
                }
        }
        free(ticks);
    }
    if (call->trace)
        fprintf(stderr, "szs call: %.1f us = setup %.1f + plan %.1f + prepare %.1f + launch %.1f + wait %.1f + wrap %.1f | kernel %.1f us | %s\n",
                profile->host_milliseconds * 1e3, call->phases[0] * 1e3, call->phases[1] * 1e3, call->phases[2] * 1e3,
                call->phases[3] * 1e3, call->phases[4] * 1e3, call->phases[5] * 1e3, kernel_ms * 1e3,
                profile->planner == 3   ? "previous plan of the same tapes, validated in the kernels"
                : profile->planner == 4 ? "planned inside the scoring launch"
                : profile->planner == 5 ? "not planned: tiny tokens, straight from the tapes"
                : profile->planner == 2 ? "device-planned, speculated"
                : profile->planner == 1 ? "device-planned"
                                        : "host-planned");
    return szs_report(sz_success_k, call->error_message, NULL);
}

/** Where do results go?  Matrices in device memory are written in place.  Plain host memory cannot be written by a
 *  kernel at all, and unified / pinned memory only across the host link, 8 scattered bytes at a time (measured on
 *  config 2: 0.90 ms instead of 0.22 ms of kernel time) - those are staged densely in HBM and copied out in one
 *  piece, unless the matrix is so small that the extra copy costs more than it saves. */
sz_status_t szs_call_place_results(szs_call_t *call) {
    szs_engine_s *engine = call->engine;
    szs_pointer_traits_t const traits = szs_classify_pointer(call->results);
    size_t const matrix_bytes = (size_t)call->q_count * call->c_count * sizeof(uint64_t);
    call->direct = traits.device_accessible && (traits.device_resident || matrix_bytes < ((size_t)256 << 10));
    call->device_results = call->results, call->device_stride = call->results_row_stride;
    if (call->direct) return sz_success_k;
    sz_status_t const status = szs_buffer_reserve(&engine->device_results, szs_memory_device_k, call->device, matrix_bytes, call->error_message);
    if (status != sz_success_k) return status;
    call->device_results = engine->device_results.pointer, call->device_stride = call->c_count;
    return sz_success_k;
}

/* ---- device-planned calls ---------------------------------------------------------------------------------------------- */

/** The refs on the device are the complete plan of exactly these tapes: remember what they were planned from. */
void szs_call_stamp_refs(szs_decision_t *remembered, void const *const data[2], void const *const offsets[2], int const wide[2],
                       szs_plan_summary_t const *summary) {
    for (int s = 0; s < 2; ++s) remembered->key_data[s] = data[s], remembered->key_offsets[s] = offsets[s], remembered->key_wide[s] = wide[s];
    remembered->summary = *summary;
    /* the shape was remembered from an earlier batch; the CELLS are this batch's (a re-used plan reports them in its profile) */
    remembered->plan.cells = remembered->symmetric ? summary->symmetric_cells : summary->side[0].symbols * summary->side[1].symbols;
    remembered->refs_current = 1;
}

#define SZS_PLAN_DEVICE_MOST_STRINGS (1u << 18) /* per side; one workgroup plans, so larger batches go to the host planner */

static int device_plannable(szs_engine_s const *engine, szs_input_t const *input) {
    if (input->kind == szs_input_sequence_k || !input->offsets || !input->data) return 0;
    if (input->count > SZS_PLAN_DEVICE_MOST_STRINGS) return 0;
    (void)engine;
    return szs_classify_pointer(input->offsets).device_accessible && szs_classify_pointer(input->data).device_accessible;
}

/** 256 bytes of device memory per engine, zeroed when allocated and never again: the two `ready` words of the launch that plans
 *  itself (dwords 0 and 32; kernels.h: szs_fused_plan_t) and the verdict words of the two-workgroup planner (dwords 48 ... 55). */
sz_status_t szs_call_reserve_device_words(szs_engine_s *engine, int device, hipStream_t stream, char const **error_message) {
    sz_status_t const status = szs_buffer_reserve(&engine->device_fused, szs_memory_device_k, device, 256, error_message);
    if (status != sz_success_k) return status;
    if (engine->fused_zeroed != engine->device_fused.pointer) {
        hipError_t const error = hipMemsetAsync(engine->device_fused.pointer, 0, 256, stream);
        if (error != hipSuccess) return szs_report_hip(error, error_message);
        engine->fused_zeroed = engine->device_fused.pointer;
    }
    return sz_success_k;
}

/* ---- host-planned calls ------------------------------------------------------------------------------------------------ */

static sz_status_t cross_host_planned(szs_call_t *call) {
    szs_engine_s *engine = call->engine;
    hipStream_t const stream = call->stream;
    int const device = call->device, symmetric = call->symmetric;
    uint32_t const q_count = call->q_count, c_count = call->c_count;
    szs_input_t const *const queries = call->queries, *const candidates = call->candidates;
    char const **error_message = call->error_message;
    size_t const most = q_count > c_count ? q_count : c_count;

    /* Pinned staging: [refs of queries][refs of candidates][offset downloads of both sides] */
    size_t const refs_bytes = ((size_t)q_count + c_count) * sizeof(szs_string_ref_t);
    size_t const offsets_bytes = ((size_t)q_count + c_count + 2) * sizeof(uint64_t);
    sz_status_t status = szs_buffer_reserve(&engine->pinned_staging, szs_memory_pinned_k, device, refs_bytes + offsets_bytes, error_message);
    if (status != sz_success_k) return status;
    status = szs_buffer_reserve(&engine->device_refs, szs_memory_device_k, device, refs_bytes, error_message);
    if (status != sz_success_k) return status;

    /* Offsets in device-only memory: both downloads are enqueued back to back and waited for ONCE. */
    void const *q_offsets = NULL, *c_offsets = NULL;
    int downloads_pending = 0;
    status = szs_prefetch_offsets(engine->pinned_staging.pointer, stream, queries, refs_bytes, &q_offsets, &downloads_pending, error_message);
    if (status == sz_success_k && !symmetric)
        status = szs_prefetch_offsets(engine->pinned_staging.pointer, stream, candidates,
                                      refs_bytes + ((size_t)q_count + 1) * sizeof(uint64_t), &c_offsets, &downloads_pending, error_message);
    if (downloads_pending) {
        hipError_t const error = hipStreamSynchronize(stream); /* also when the second enqueue failed */
        if (status == sz_success_k && error != hipSuccess) return szs_report_hip(error, error_message);
    }
    if (status != sz_success_k) return status;
    szs_call_phase(call, 0); /* checks, buffers, offsets download + its synchronisation */

    /* Host scratch: [q addresses][c addresses][q lengths][c lengths] */
    size_t const fixed_bytes = ((size_t)q_count + c_count) * (sizeof(uint64_t) + sizeof(uint32_t));
    status = szs_buffer_reserve(&engine->host_lengths, szs_memory_host_k, 0, fixed_bytes, error_message);
    if (status != sz_success_k) return status;
    uint64_t *q_addresses = (uint64_t *)engine->host_lengths.pointer;
    uint64_t *c_addresses = q_addresses + q_count;
    uint32_t *q_lengths = (uint32_t *)(c_addresses + c_count);
    uint32_t *c_lengths = q_lengths + q_count;

    uint64_t query_bytes = 0, candidate_bytes = 0;
    int const may_stage = szs_tuning_get(szs_knob_cpu_requests_k) == 1;
    int stage_queries = 0, stage_candidates = 0;
    status = szs_gather_strings(queries, q_offsets, q_addresses, q_lengths, &query_bytes, may_stage ? &stage_queries : NULL, error_message);
    if (status != sz_success_k) return status;
    if (!symmetric) {
        status = szs_gather_strings(candidates, c_offsets, c_addresses, c_lengths, &candidate_bytes, may_stage ? &stage_candidates : NULL,
                                    error_message);
        if (status != sz_success_k) return status;
    }
    if (stage_queries || stage_candidates) { /* host-only strings of a caller that asked for a CPU engine */
        size_t const tape_bytes = (stage_queries ? query_bytes : 0) + (stage_candidates ? candidate_bytes : 0) + 16;
        status = szs_buffer_reserve(&engine->pinned_tape, szs_memory_pinned_k, device, tape_bytes, error_message);
        if (status == sz_success_k) status = szs_buffer_reserve(&engine->device_tape, szs_memory_device_k, device, tape_bytes, error_message);
        if (status == sz_success_k && stage_queries)
            status = stage_host_strings(engine, stream, q_addresses, q_lengths, q_count, query_bytes, 0, error_message);
        if (status == sz_success_k && stage_candidates)
            status = stage_host_strings(engine, stream, c_addresses, c_lengths, c_count, candidate_bytes,
                                        stage_queries ? query_bytes : 0, error_message);
        if (status != sz_success_k) {
            (void)hipStreamSynchronize(stream);
            return status;
        }
    }
    if (symmetric) {
        memcpy(c_addresses, q_addresses, (size_t)q_count * sizeof(uint64_t));
        memcpy(c_lengths, q_lengths, (size_t)q_count * sizeof(uint32_t));
        candidate_bytes = query_bytes;
    }

    /* Codepoint-level engine: transcode every string to UTF-32 ONCE (hip/utf8.hip), then plan and score on runes.  When
     * no string holds a byte >= 0x80 the corpus is ASCII and the byte kernels compute the same distances - the
     * reference takes the same shortcut pair by pair (serial.hpp:2809-2813). */
    int runes = 0;
    uint32_t alphabet = 0;
    if (engine->family == szs_family_levenshtein_utf8_k) {
        status = transcode_to_runes(engine, device, stream, symmetric, q_addresses, q_lengths, q_count, c_addresses, c_lengths, c_count,
                                    &runes, &alphabet, error_message);
        if (status != sz_success_k) return status;
    }

    int const use_myers = engine->is_unit_cost && (engine->family == szs_family_levenshtein_k || engine->family == szs_family_levenshtein_utf8_k);
    unsigned const myers_words = !use_myers ? 0 : SZS_MYERS_MAX_WORDS; /* bytes and codepoints alike: up to 2048 symbols */
    szs_side_stats_t q_stats, c_stats;
    uint32_t q_variants[SZS_PLAN_VARIANTS], c_variants[SZS_PLAN_VARIANTS];
    szs_side_stats(q_lengths, q_count, myers_words, &q_stats, q_variants);
    szs_side_stats(c_lengths, c_count, myers_words, &c_stats, c_variants);
    uint64_t cells = q_stats.symbols * c_stats.symbols;
    if (symmetric) { /* lower triangle incl. diagonal: sum_i len_i * sum_{j <= i} len_j */
        uint64_t prefix = 0;
        cells = 0;
        for (uint32_t i = 0; i < q_count; ++i) prefix += q_lengths[i], cells += (uint64_t)q_lengths[i] * prefix;
    }
    /* The planner's own scratch (sort keys + counting bins), grow-only like everything else: no allocation per call. */
    uint32_t const longest = q_stats.longest > c_stats.longest ? q_stats.longest : c_stats.longest;
    size_t const keys_bytes = (most * sizeof(uint32_t) + 15) & ~(size_t)15;
    status = szs_buffer_reserve(&engine->host_scratch, szs_memory_host_k, 0, keys_bytes + szs_plan_scratch_bytes((uint32_t)most, longest), error_message);
    if (status != sz_success_k) return status;
    uint32_t *const keys = (uint32_t *)engine->host_scratch.pointer;
    void *const scratch = (char *)engine->host_scratch.pointer + keys_bytes;

    status = szs_call_place_results(call);
    if (status != sz_success_k) return status;

    for (int attempt = 0; attempt < 2; ++attempt) { /* second round: a stalled band chain is re-run on the lanes tier */
        szs_decision_t d;
        status = szs_call_decide(engine, symmetric, runes, attempt > 0, &q_stats, &c_stats, q_variants, c_variants, NULL /* ranks: after the sort below */,
                        cells, &d, error_message);
        if (status != sz_success_k) return status;
        d.alphabet = alphabet;
        szs_call_decide_queue(engine, &d, NULL); /* planned below, once the lengths are sorted */
        /* kernel roles */
        uint64_t *const kq_addresses = d.transposed ? c_addresses : q_addresses, *const kc_addresses = d.transposed ? q_addresses : c_addresses;
        uint32_t *const kq_lengths = d.transposed ? c_lengths : q_lengths, *const kc_lengths = d.transposed ? q_lengths : c_lengths;

        /* Plan straight into the pinned staging area, then ship both ref arrays in one copy. */
        szs_string_ref_t *host_query_refs = (szs_string_ref_t *)engine->pinned_staging.pointer;
        szs_string_ref_t *host_candidate_refs = host_query_refs + d.kq_count;
        szs_plan_t sorted;
        szs_plan_build(myers_words, symmetric, kq_addresses, kq_lengths, d.kq_count, kc_addresses, kc_lengths, d.kc_count, host_query_refs,
                       host_candidate_refs, keys, scratch, &sorted);
        if (d.use_queue) { /* the queue of the one-launch kernel is ordered by the sorted lengths (kernel roles already) */
            d.plan.has_ranks = 1;
            memcpy(d.plan.rank_lengths, sorted.rank_lengths, sizeof(d.plan.rank_lengths));
            szs_plan_queue(&d.plan, d.kq_count, d.kc_count, d.runes ? d.alphabet : 0u, szs_hip_levenshtein_myers_queue_table_bytes(d.runes), &d.queue);
            if (!d.queue.items_total) d.use_queue = 0;
        }
        szs_call_phase(call, 1); /* gathering strings, transcoding, orientation, planning */

        status = szs_call_prepare(engine, &d, device, stream, error_message);
        if (status != sz_success_k) return status;
        szs_string_ref_t *device_query_refs = (szs_string_ref_t *)engine->device_refs.pointer;
        szs_string_ref_t *device_candidate_refs = device_query_refs + d.kq_count;
        hipError_t error = hipMemcpyAsync(device_query_refs, host_query_refs, refs_bytes, hipMemcpyHostToDevice, stream);
        szs_call_phase(call, 2); /* ref upload enqueued, result placement, workspaces */

        /* ---- launches, bracketed by the engine's event pair on the scope's stream ---- */
        uint32_t launches = 0, cell_bits = 0;
        sz_status_t enqueue_status = sz_success_k;
        if (error == hipSuccess) error = hipEventRecord(engine->event_start, stream);
        if (error == hipSuccess)
            error = szs_call_enqueue(engine, &d, device, device_query_refs, device_candidate_refs, call->device_results, call->device_stride, stream,
                            NULL, &launches, &cell_bits, &enqueue_status, error_message);
        int stalled = 0;
        engine->last_profile.planner = 0;
        status = szs_call_finish(call, &d, error, enqueue_status, launches, cell_bits, query_bytes, candidate_bytes, &stalled);
        if (status != sz_success_k || !stalled) return status;
    }
    return szs_report(sz_status_unknown_k, error_message, "Systolic pipeline stalled");
}

sz_status_t szs_engine_cross(szs_engine_s *engine, szs_scope_s *scope, szs_input_t const *queries,
                             szs_input_t const *candidates, void *results, size_t results_row_stride,
                             char const **error_message) {
    szs_call_t call;
    memset(&call, 0, sizeof(call));
    call.started = call.phase_started = now_milliseconds();
    call.trace = szs_tuning_get(szs_knob_trace_k) > 0; /* per-phase host times of every call on stderr (a measuring aid) */
    if (!engine || engine->magic != SZS_ENGINE_MAGIC)
        return szs_report(sz_status_unknown_k, error_message, "Engine must be initialized");
    if (!queries) return szs_report(sz_status_unknown_k, error_message, "Queries must not be null");

    int device = 0;
    hipStream_t stream = NULL;
    sz_status_t status = szs_scope_bind_gpu(scope, &device, &stream, error_message);
    if (status != sz_success_k) return status;

    int const symmetric = candidates == NULL;
    size_t const queries_count = queries->count;
    size_t const candidates_count = symmetric ? queries_count : candidates->count;
    engine->cells_before = engine->last_profile.cells;
    memset(&engine->last_profile, 0, sizeof(engine->last_profile));
    /* The dense byte alphabet of a non-unit Levenshtein engine belongs to ONE call: only the device-planned path scans the
     * tapes and fills it, and szs_call_decide() / fill_cost_model() read it on every path - a host-planned call after a device-planned
     * one must not score with the previous batch's byte-to-class map (bytes that batch lacked would all share class 0). */
    engine->uniform_classes = 0;
    if (!queries_count || !candidates_count) return szs_report(sz_success_k, error_message, NULL); /* cuda.cuh:4257 */
    if (queries_count > 0xFFFFFFFFull || candidates_count > 0xFFFFFFFFull)
        return szs_report(sz_overflow_risk_k, error_message, NULL);
    if (!results) return szs_report(sz_status_unknown_k, error_message, "Results must not be null");
    if (results_row_stride < candidates_count) return szs_report(sz_unexpected_dimensions_k, error_message, NULL);

    if (engine->device != device) { /* scratch follows the device of the call */
        if (engine->device >= 0) {
            (void)hipSetDevice(engine->device);
            release_device_state(engine);
            (void)hipSetDevice(device);
        }
        engine->device = device;
    }
    if (engine->events_device != device) {
        hipError_t error = hipEventCreate(&engine->event_start);
        if (error == hipSuccess) error = hipEventCreate(&engine->event_stop);
        if (error != hipSuccess) return szs_report_hip(error, error_message);
        engine->events_device = device;
    }
    /* pinned: the device planner's summary, and behind it the stall flag of the chained tiers */
    status = szs_buffer_reserve(&engine->pinned_summary, szs_memory_pinned_k, device, 2048, error_message); /* [1024, 2048): the reports of a fused launch */
    if (status != sz_success_k) return status;

    call.engine = engine, call.device = device, call.stream = stream;
    call.queries = queries, call.candidates = candidates, call.symmetric = symmetric;
    call.q_count = (uint32_t)queries_count, call.c_count = (uint32_t)candidates_count;
    call.results = results, call.results_row_stride = results_row_stride, call.error_message = error_message;

    int const planner = szs_tuning_get(szs_knob_planner_k);
    engine->queue_refused = 0;
    for (int round = 0; round < 2; ++round) { /* second round: the one-launch kernel met a query it has no table for (see enqueue) */
        ranges_begin(&call);
        engine->queue_unfit_sequence = 0;
        status = SZS_NOT_DEVICE_PLANNABLE;
        if (planner != 0 && device_plannable(engine, queries) && (symmetric || device_plannable(engine, candidates))) {
            status = engine->family == szs_family_levenshtein_utf8_k ? szs_cross_device_planned_runes(&call) : SZS_RUNES_ARE_BYTES;
            if (status == SZS_RUNES_ARE_BYTES) {
                if (call.ranges) ranges_end(&call), ranges_begin(&call);
                status = szs_cross_device_planned(&call);
            }
        }
        if (status == SZS_NOT_DEVICE_PLANNABLE) {
            if (call.ranges) ranges_end(&call), ranges_begin(&call); /* the phases start over */
            status = cross_host_planned(&call);
        }
        ranges_end(&call);
        uint32_t const raised = *(uint32_t const volatile *)((char const *)engine->pinned_summary.pointer + 960);
        if (status != sz_success_k || !engine->queue_unfit_sequence || raised != engine->queue_unfit_sequence) break;
        engine->queue_refused = 1; /* the per-width launches score every cell again */
        if (engine->remembered) engine->remembered->valid = 0;
    }
    engine->queue_refused = 0;
    return status;
}
