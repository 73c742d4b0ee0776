/*
 *  dispatch.c - one engine call, start to finish: normalise inputs -> plan -> upload refs -> launch -> synchronise.
 *
 *  ROCm counterpart of the reference's `cross_()` / `run_trampoline_()` (cuda.cuh:4247-4417,4435-4741) and
 *  `cuda_weighted_cross_()` (cuda.cuh:5913).  Same observable behaviour - synchronous call, results in the caller's
 *  matrix, device-accessibility checks on the strings, staged copy when `results` is not device-visible - with a
 *  different mechanism (see plan.c).  Nothing here computes a score on the CPU.
 */
#include "szs_internal.h"

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
                               uint64_t *total_bytes, char const **error_message) {
    size_t const count = input->count;
    *total_bytes = 0;
    if (input->kind == szs_input_sequence_k) {
        sz_sequence_t const *sequence = input->sequence;
        int checked = 0;
        for (size_t i = 0; i < count; ++i) {
            char const *start = sequence->get_start(sequence->handle, i);
            size_t const length = sequence->get_length(sequence->handle, i);
            if (length > 0xFFFFFFFFull) return szs_report(sz_overflow_risk_k, error_message, NULL);
            if (length && !checked) { /* like the reference, vet one representative string (cuda.cuh:4268-4272) */
                if (!szs_classify_pointer(start).device_accessible)
                    return szs_report(sz_device_memory_mismatch_k, error_message, NULL);
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
    if (*total_bytes && !szs_classify_pointer(input->data).device_accessible)
        return szs_report(sz_device_memory_mismatch_k, error_message, NULL);
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
                                      char const **error_message) {
    size_t const strings = (size_t)q_count + (symmetric ? 0 : c_count);
    /* staging layout, host and device alike: [refs][rune starts][rune counts][flag] */
    size_t const refs_at = 0, starts_at = refs_at + strings * sizeof(szs_string_ref_t);
    size_t const counts_at = starts_at + strings * sizeof(uint64_t), flag_at = counts_at + strings * sizeof(uint32_t);
    size_t const staging_bytes = flag_at + sizeof(uint32_t);
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
        starts[i] = total, total += refs[i].length;
    }
    *(uint32_t *)(host + flag_at) = 0;
    status = szs_buffer_reserve(&engine->device_runes, szs_memory_device_k, device, (total + 1) * sizeof(uint32_t), error_message);
    if (status != sz_success_k) return status;

    hipError_t error = hipMemcpyAsync(remote, host, counts_at, hipMemcpyHostToDevice, stream);
    if (error == hipSuccess) error = hipMemsetAsync(remote + flag_at, 0, sizeof(uint32_t), stream);
    if (error != hipSuccess) return szs_report_hip(error, error_message);
    int const launch_error = szs_hip_utf8_transcode(
        (szs_string_ref_t const *)(remote + refs_at), (uint32_t)strings, (uint64_t const *)(remote + starts_at),
        (uint32_t *)engine->device_runes.pointer, (uint32_t *)(remote + counts_at), (uint32_t *)(remote + flag_at), stream);
    if (launch_error) return szs_report_hip((hipError_t)launch_error, error_message);
    error = hipMemcpyAsync(host + counts_at, remote + counts_at, staging_bytes - counts_at, hipMemcpyDeviceToHost, stream);
    if (error == hipSuccess) error = hipStreamSynchronize(stream);
    if (error != hipSuccess) return szs_report_hip(error, error_message);

    *runes = *(uint32_t const *)(host + flag_at) != 0;
    if (!*runes) return sz_success_k;
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
    szs_buffer_release(&engine->device_runes);
    szs_buffer_release(&engine->device_transcode);
    szs_buffer_release(&engine->pinned_transcode);
    if (engine->events_device >= 0) {
        (void)hipEventDestroy(engine->event_start);
        (void)hipEventDestroy(engine->event_stop);
        engine->events_device = -1;
    }
    engine->model_uploaded_device = -1;
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
}

sz_status_t szs_engine_cross(szs_engine_s *engine, szs_scope_s *scope, szs_input_t const *queries,
                             szs_input_t const *candidates, void *results, size_t results_row_stride,
                             char const **error_message) {
    double const call_started = now_milliseconds();
    static int trace = -1; /* SZS_ROCM_TRACE=1: per-phase host times of every call on stderr (a measuring aid) */
    if (trace < 0) trace = getenv("SZS_ROCM_TRACE") != NULL;
    double phase_started = call_started, phases[6] = {0, 0, 0, 0, 0, 0};
#define SZS_PHASE(INDEX)                                                                                               \
    do {                                                                                                               \
        if (trace) {                                                                                                   \
            double const now = now_milliseconds();                                                                     \
            phases[INDEX] += now - phase_started, phase_started = now;                                                 \
        }                                                                                                              \
    } while (0)
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
    memset(&engine->last_profile, 0, sizeof(engine->last_profile));
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

    uint32_t const q_count = (uint32_t)queries_count, c_count = (uint32_t)candidates_count;
    size_t const most = q_count > c_count ? q_count : c_count;

    /* Host scratch: [q addresses][c addresses][q lengths][c lengths][sort keys] */
    size_t const host_bytes = ((size_t)q_count + c_count) * (sizeof(uint64_t) + sizeof(uint32_t)) + most * sizeof(uint32_t);
    status = szs_buffer_reserve(&engine->host_lengths, szs_memory_host_k, 0, host_bytes, error_message);
    if (status != sz_success_k) return status;
    uint64_t *q_addresses = (uint64_t *)engine->host_lengths.pointer;
    uint64_t *c_addresses = q_addresses + q_count;
    uint32_t *q_lengths = (uint32_t *)(c_addresses + c_count);
    uint32_t *c_lengths = q_lengths + q_count;
    uint32_t *keys = c_lengths + c_count;

    /* Pinned staging: [refs of queries][refs of candidates][offset downloads of both sides] */
    size_t const refs_bytes = ((size_t)q_count + c_count) * sizeof(szs_string_ref_t);
    size_t const offsets_bytes = ((size_t)q_count + c_count + 2) * sizeof(uint64_t);
    status = szs_buffer_reserve(&engine->pinned_staging, szs_memory_pinned_k, device, refs_bytes + offsets_bytes,
                                error_message);
    if (status != sz_success_k) return status;
    status = szs_buffer_reserve(&engine->device_refs, szs_memory_device_k, device, refs_bytes, error_message);
    if (status != sz_success_k) return status;

    /* Offsets in device-only memory: both downloads are enqueued back to back and waited for ONCE. */
    void const *q_offsets = NULL, *c_offsets = NULL;
    int downloads_pending = 0;
    status = szs_prefetch_offsets(engine->pinned_staging.pointer, stream, queries, refs_bytes, &q_offsets, &downloads_pending,
                                  error_message);
    if (status != sz_success_k) return status;
    if (!symmetric) {
        status = szs_prefetch_offsets(engine->pinned_staging.pointer, stream, candidates,
                                      refs_bytes + ((size_t)q_count + 1) * sizeof(uint64_t),
                                  &c_offsets, &downloads_pending, error_message);
        if (status != sz_success_k) return status;
    }
    if (downloads_pending) {
        hipError_t const error = hipStreamSynchronize(stream);
        if (error != hipSuccess) return szs_report_hip(error, error_message);
    }

    SZS_PHASE(0); /* checks, buffers, offsets download + its synchronisation */
    uint64_t query_bytes = 0, candidate_bytes = 0;
    status = szs_gather_strings(queries, q_offsets, q_addresses, q_lengths, &query_bytes, error_message);
    if (status != sz_success_k) return status;
    if (symmetric) {
        memcpy(c_addresses, q_addresses, (size_t)q_count * sizeof(uint64_t));
        memcpy(c_lengths, q_lengths, (size_t)q_count * sizeof(uint32_t));
        candidate_bytes = query_bytes;
    }
    else {
        status = szs_gather_strings(candidates, c_offsets, c_addresses, c_lengths, &candidate_bytes, error_message);
        if (status != sz_success_k) return status;
    }

    /* Codepoint-level engine: transcode every string to UTF-32 ONCE (hip/utf8.hip), then plan and score on runes.  When
     * no string holds a byte >= 0x80 the corpus is ASCII and the byte kernels compute the same distances - the
     * reference takes the same shortcut pair by pair (serial.hpp:2809-2813). */
    int runes = 0;
    if (engine->family == szs_family_levenshtein_utf8_k) {
        status = transcode_to_runes(engine, device, stream, symmetric, q_addresses, q_lengths, q_count, c_addresses,
                                    c_lengths, c_count, &runes, error_message);
        if (status != sz_success_k) return status;
    }

    int const use_myers = engine->is_unit_cost && (engine->family == szs_family_levenshtein_k ||
                                                   engine->family == szs_family_levenshtein_utf8_k);
    int const maximise = engine->family == szs_family_needleman_wunsch_k || engine->family == szs_family_smith_waterman_k;
    unsigned const myers_words = !use_myers ? 0 : SZS_MYERS_MAX_WORDS; /* bytes and codepoints alike: up to 2048 symbols */

    /* ---- orientation and tier.  Every kernel puts ONE side on workgroups / band chains (its "queries") and the other
     * on lanes / columns (its "candidates"); which real side plays which role is free - gap costs apply to both strings
     * alike and a swapped class table is its transpose - so a cycle model of both tiers (plan.c) is evaluated for both
     * orientations and the cheaper one runs.  1024 queries x 1 candidate thus become 1 workgroup row of 1024 lanes
     * instead of 1024 workgroups with one live lane each.  Symmetric calls have nothing to swap. */
    uint32_t q_longest = 0, c_longest = 0;
    for (uint32_t i = 0; i < q_count; ++i) q_longest = q_lengths[i] > q_longest ? q_lengths[i] : q_longest;
    for (uint32_t i = 0; i < c_count; ++i) c_longest = c_lengths[i] > c_longest ? c_lengths[i] : c_longest;
    int tier = SZS_TIER_LANES, transposed = 0;
    /* bit-parallel at any length for bytes (2048-row strips beyond 64 words), up to 2048 symbols for codepoints */
    int const banded = use_myers && !runes;
    szs_plan_orient(banded ? 0xFFFFFFFFu : myers_words * 32, use_myers && !runes, !engine->is_linear, !maximise, symmetric, q_lengths, q_count,
                    c_lengths, c_count, szs_hip_systolic_band_rows(), &tier, &transposed);
    char const *const forced_tier = getenv("SZS_ROCM_TIER"); /* `systolic` on a unit-cost engine means the DP recurrences */
    if (tier == SZS_TIER_MYERS_CHAIN && forced_tier && forced_tier[0] == 's') tier = SZS_TIER_SYSTOLIC;
    /* kernel roles */
    uint64_t *const kq_addresses = transposed ? c_addresses : q_addresses, *const kc_addresses = transposed ? q_addresses : c_addresses;
    uint32_t *const kq_lengths = transposed ? c_lengths : q_lengths, *const kc_lengths = transposed ? q_lengths : c_lengths;
    uint32_t const kq_count = transposed ? c_count : q_count, kc_count = transposed ? q_count : c_count;
    int const layout = (symmetric ? SZS_LAYOUT_SYMMETRIC : 0) | (transposed ? SZS_LAYOUT_TRANSPOSED : 0);

    /* Plan straight into the pinned staging area, then ship both ref arrays in one copy. */
    szs_string_ref_t *host_query_refs = (szs_string_ref_t *)engine->pinned_staging.pointer;
    szs_string_ref_t *host_candidate_refs = host_query_refs + kq_count;
    szs_plan_t plan;
    szs_plan_build(myers_words, symmetric, kq_addresses, kq_lengths, kq_count, kc_addresses, kc_lengths, kc_count,
                   host_query_refs, host_candidate_refs, keys, &plan);

    /* Cell width: this build scores weighted cells in 32 bits, so refuse what the reference would widen to 64 bits
     * (reach rule, serial.hpp:135-162,370-386). */
    uint64_t const span = maximise ? (uint64_t)plan.longest_query + plan.longest_candidate
                                   : (plan.longest_query > plan.longest_candidate ? plan.longest_query : plan.longest_candidate);
    uint64_t const reach = (span + (engine->is_linear ? 1 : 3)) * (engine->magnitude ? engine->magnitude : 1);
    if (reach >= 0x7FFFFFF0ull) return szs_report(sz_overflow_risk_k, error_message, NULL);

    SZS_PHASE(1); /* gathering strings, transcoding, orientation, planning */
    szs_string_ref_t *device_query_refs = (szs_string_ref_t *)engine->device_refs.pointer;
    szs_string_ref_t *device_candidate_refs = device_query_refs + kq_count;
    hipError_t error = hipMemcpyAsync(device_query_refs, host_query_refs, refs_bytes, hipMemcpyHostToDevice, stream);
    if (error != hipSuccess) return szs_report_hip(error, error_message);

    /* Where do results go?  Matrices in device memory are written in place.  Plain host memory cannot be written by a
     * kernel at all, and unified / pinned memory only across the host link, 8 scattered bytes at a time (measured on
     * config 2: 0.90 ms instead of 0.22 ms of kernel time) - those are staged densely in HBM and copied out in one
     * piece, unless the matrix is so small that the extra copy costs more than it saves. */
    szs_pointer_traits_t const results_traits = szs_classify_pointer(results);
    int const direct = results_traits.device_accessible &&
                       (results_traits.device_resident || (size_t)q_count * c_count * sizeof(uint64_t) < ((size_t)256 << 10));
    void *device_results = results;
    size_t device_stride = results_row_stride;
    if (!direct) {
        status = szs_buffer_reserve(&engine->device_results, szs_memory_device_k, device,
                                    (size_t)q_count * c_count * sizeof(uint64_t), error_message);
        if (status != sz_success_k) return status;
        device_results = engine->device_results.pointer, device_stride = c_count;
    }

    int const objective = engine->family == szs_family_needleman_wunsch_k   ? szs_objective_global_k
                          : engine->family == szs_family_smith_waterman_k
                              ? (engine->open <= 0 && engine->extend <= 0 ? szs_objective_local_saturating_k : szs_objective_local_k)
                          : runes                                         ? szs_objective_distance_runes_k
                                                                          : szs_objective_distance_k;
    /* 16-bit strip boundaries when every parked value provably fits: global scores are bounded by the reach, saturating
     * local ones by (shorter side) x (largest cost) - the reference narrows its cells by the same kind of bound. */
    uint64_t const shorter_side = plan.longest_query < plan.longest_candidate ? plan.longest_query : plan.longest_candidate;
    int const narrow = objective == szs_objective_global_k ? reach < 32000
                       : objective == szs_objective_local_saturating_k
                           ? (shorter_side + 3) * (engine->magnitude ? engine->magnitude : 1) < 32000
                           : 0;

    /* Two cells per VALU operation (hip/weighted_packed.hip) when EVERY DP value fits 16 bits, not just the parked ones:
     * the same two bounds cover all cells and tracks - the reach is a bound on any sum of (rows + columns + 3) costs. */
    uint32_t classes = 0;
    if (maximise)
        for (int i = 0; i < 256; ++i) classes = engine->byte_to_class[i] >= classes ? (uint32_t)engine->byte_to_class[i] + 1 : classes;
    char const *const forced_packed = getenv("SZS_ROCM_PACKED"); /* testing aid: 0 pins the 32-bit kernel */
    int const packed = maximise && narrow && classes <= 32 && !(forced_packed && forced_packed[0] == '0');
    int const packed_local = objective == szs_objective_local_saturating_k;

    /* The systolic tier scores every engine family with its weighted recurrences, so it needs the cost model and its
     * own workspace; a job with too many pairs in flight for that workspace stays on the lanes tier. */
    size_t systolic_control_bytes = 0, systolic_parked_bytes = 0;
    if (tier == SZS_TIER_SYSTOLIC &&
        (!szs_hip_systolic_workspace_bytes(!engine->is_linear, kq_count, kc_count, plan.longest_query, plan.longest_candidate,
                                           &systolic_control_bytes, &systolic_parked_bytes) ||
         systolic_control_bytes + systolic_parked_bytes > ((size_t)32 << 30)))
        tier = SZS_TIER_LANES;
    if (tier == SZS_TIER_MYERS_CHAIN &&
        (!szs_hip_myers_chain_workspace_bytes(kq_count, kc_count, plan.longest_query, plan.longest_candidate,
                                              &systolic_control_bytes, &systolic_parked_bytes) ||
         systolic_control_bytes + systolic_parked_bytes > ((size_t)32 << 30)))
        tier = SZS_TIER_LANES;
    int const chained = tier == SZS_TIER_SYSTOLIC || tier == SZS_TIER_MYERS_CHAIN;
    if (chained) {
        /* The control block is zeroed when it is (re)allocated and never again: its words carry the epoch of the launch
         * that wrote them, so a launch neither needs nor waits for a fill (hip/kernels.h). */
        int const fresh = !engine->device_systolic.pointer || engine->device_systolic.capacity < systolic_control_bytes ||
                          engine->device_systolic.device != device; /* the reserve below will (re)allocate */
        status = szs_buffer_reserve(&engine->device_systolic, szs_memory_device_k, device, systolic_control_bytes, error_message);
        if (status != sz_success_k) return status;
        if (fresh || engine->systolic_epoch >= 0xFFFFFFF0u) {
            error = hipMemsetAsync(engine->device_systolic.pointer, 0, engine->device_systolic.capacity, stream);
            if (error == hipSuccess) error = hipStreamSynchronize(stream);
            if (error != hipSuccess) return szs_report_hip(error, error_message);
            engine->systolic_epoch = 0;
        }
        engine->systolic_epoch++;
    }

    /* Weighted kernels need the cost model and a strip-boundary workspace on the device. */
    int needs_weighted = !use_myers || tier == SZS_TIER_SYSTOLIC || runes; /* runes: the fallback of the long rune kernels */
    if (tier == SZS_TIER_MYERS_CHAIN) { /* no cost model; its parked deltas live in the boundary buffer like the systolic rows */
        status = szs_buffer_reserve(&engine->device_boundary, szs_memory_device_k, device, systolic_parked_bytes, error_message);
        if (status != sz_success_k) return status;
    }
    for (unsigned g = 0; tier != SZS_TIER_MYERS_CHAIN && g < plan.groups_count; ++g) needs_weighted |= plan.groups[g].variant == 0;
    if (needs_weighted) {
        if (engine->model_uploaded_device != device || engine->model_uploaded_transposed != transposed) {
            status = szs_buffer_reserve(&engine->device_model, szs_memory_device_k, device, sizeof(szs_cost_model_t),
                                        error_message);
            if (status != sz_success_k) return status;
            szs_cost_model_t model;
            fill_cost_model(engine, transposed, &model);
            error = hipMemcpyAsync(engine->device_model.pointer, &model, sizeof(model), hipMemcpyHostToDevice, stream);
            if (error == hipSuccess) error = hipStreamSynchronize(stream); /* `model` lives on this stack frame */
            if (error != hipSuccess) return szs_report_hip(error, error_message);
            engine->model_uploaded_device = device, engine->model_uploaded_transposed = transposed;
        }
        size_t boundary_bytes =
            tier == SZS_TIER_SYSTOLIC
                ? systolic_parked_bytes
                : packed ? szs_hip_weighted_packed_boundary_bytes(packed_local, !engine->is_linear, classes, kq_count, kc_count,
                                                                  plan.longest_candidate)
                         : szs_hip_weighted_boundary_bytes(objective, !engine->is_linear, narrow, kq_count, kc_count, plan.longest_candidate);
        if (banded && tier == SZS_TIER_LANES) { /* queries beyond 64 words: the strip kernel's parked deltas instead */
            size_t const banded_bytes = szs_hip_levenshtein_myers_banded_bytes(kq_count, kc_count, plan.longest_candidate);
            boundary_bytes = banded_bytes > boundary_bytes ? banded_bytes : boundary_bytes;
        }
        status = szs_buffer_reserve(&engine->device_boundary, szs_memory_device_k, device, boundary_bytes, error_message);
        if (status != sz_success_k) return status;
    }

    SZS_PHASE(2); /* ref upload enqueued, result placement, workspaces */
    /* ---- launches, bracketed by the engine's event pair on the scope's stream ---- */
    error = hipEventRecord(engine->event_start, stream);
    if (error != hipSuccess) return szs_report_hip(error, error_message);
    uint32_t launches = 0, cell_bits = 0;
    if (tier == SZS_TIER_SYSTOLIC) { /* one launch for the whole cross-product, whatever the planner's groups */
        int const launch_error = szs_hip_systolic_scores(
            objective, !engine->is_linear, (szs_cost_model_t const *)engine->device_model.pointer,
            device_query_refs, kq_count, device_candidate_refs, kc_count, plan.longest_query, plan.longest_candidate,
            (int64_t *)device_results, device_stride, layout, engine->device_systolic.pointer,
            engine->device_boundary.pointer, engine->systolic_epoch, stream);
        if (launch_error) return szs_report_hip((hipError_t)launch_error, error_message);
        ++launches, cell_bits = 32;
    }
    if (tier == SZS_TIER_MYERS_CHAIN) {
        int const launch_error = szs_hip_myers_chain(device_query_refs, kq_count, device_candidate_refs, kc_count,
                                                     plan.longest_query, plan.longest_candidate, (uint64_t *)device_results,
                                                     device_stride, layout, engine->device_systolic.pointer,
                                                     engine->device_boundary.pointer, engine->systolic_epoch, stream);
        if (launch_error) return szs_report_hip((hipError_t)launch_error, error_message);
        ++launches;
    }
    for (unsigned g = 0; tier == SZS_TIER_LANES && g < plan.groups_count; ++g) {
        szs_plan_group_t const *group = &plan.groups[g];
        int launch_error;
        if (group->variant && runes) {
            launch_error = group->variant == SZS_MYERS_SHORT_WORDS
                               ? szs_hip_levenshtein_myers_runes(device_query_refs + group->first, group->count,
                                                                 device_candidate_refs, kc_count, (uint64_t *)device_results,
                                                                 device_stride, layout, stream)
                               : szs_hip_levenshtein_myers_runes_long(group->variant, device_query_refs + group->first,
                                                                      group->count, device_candidate_refs, kc_count,
                                                                      (uint64_t *)device_results, device_stride, layout, stream);
            if (launch_error == (int)hipErrorNotSupported) { /* no LDS for the rune table: the rune-keyed DP kernel */
                cell_bits = 32;
                launch_error = szs_hip_weighted_scores(objective, !engine->is_linear, 0,
                                                       (szs_cost_model_t const *)engine->device_model.pointer,
                                                       device_query_refs + group->first, group->count,
                                                       device_candidate_refs, kc_count, plan.longest_candidate,
                                                       (int64_t *)device_results, device_stride, layout,
                                                       engine->device_boundary.pointer, stream);
            }
        }
        else if (group->variant)
            launch_error = szs_hip_levenshtein_myers(group->variant, device_query_refs + group->first, group->count,
                                                     device_candidate_refs, kc_count, (uint64_t *)device_results,
                                                     device_stride, layout, stream);
        else if (banded)
            launch_error = szs_hip_levenshtein_myers_banded(device_query_refs + group->first, group->count, device_candidate_refs,
                                                            kc_count, plan.longest_candidate, (uint64_t *)device_results,
                                                            device_stride, layout, engine->device_boundary.pointer, stream);
        else if (packed) {
            cell_bits = 16;
            launch_error = szs_hip_weighted_packed_scores(packed_local, !engine->is_linear, classes,
                                                          (szs_cost_model_t const *)engine->device_model.pointer,
                                                          device_query_refs + group->first, group->count,
                                                          device_candidate_refs, kc_count, plan.longest_candidate,
                                                          (int64_t *)device_results, device_stride, layout,
                                                          engine->device_boundary.pointer, stream);
        }
        else {
            cell_bits = 32;
            launch_error = szs_hip_weighted_scores(objective, !engine->is_linear, narrow,
                                                   (szs_cost_model_t const *)engine->device_model.pointer,
                                                   device_query_refs + group->first, group->count,
                                                   device_candidate_refs, kc_count, plan.longest_candidate,
                                                   (int64_t *)device_results, device_stride, layout,
                                                   engine->device_boundary.pointer, stream);
        }
        if (launch_error) return szs_report_hip((hipError_t)launch_error, error_message);
        ++launches;
    }
    error = hipEventRecord(engine->event_stop, stream);
    if (error != hipSuccess) return szs_report_hip(error, error_message);
    uint64_t *const stall_flag = (uint64_t *)((char *)engine->pinned_staging.pointer + refs_bytes); /* offsets area: done with */
    *stall_flag = 0;
    if (chained) {
        error = hipMemcpyAsync(stall_flag, (char *)engine->device_systolic.pointer + 8, sizeof(uint64_t), hipMemcpyDeviceToHost, stream);
        if (error != hipSuccess) return szs_report_hip(error, error_message);
    }

    if (!direct) /* one strided copy back into the caller's host matrix (reference: cuMemcpy2DAsync, cuda.cuh:2205-2215) */
        error = hipMemcpy2DAsync(results, results_row_stride * sizeof(uint64_t), device_results,
                                 device_stride * sizeof(uint64_t), (size_t)c_count * sizeof(uint64_t), q_count,
                                 hipMemcpyDefault, stream);
    SZS_PHASE(3); /* launches enqueued */
    if (error == hipSuccess) error = hipStreamSynchronize(stream); /* the call is synchronous, like the reference's */
    SZS_PHASE(4); /* waiting for the device */
    if (error != hipSuccess) return szs_report_hip(error, error_message);
    if (chained && *stall_flag == (((uint64_t)engine->systolic_epoch << 32) | 1))
        return szs_report(sz_status_unknown_k, error_message, "Systolic pipeline stalled");

    float kernel_ms = 0;
    (void)hipEventElapsedTime(&kernel_ms, engine->event_start, engine->event_stop);
    szs_rocm_call_profile_t *profile = &engine->last_profile;
    uint64_t const pairs = symmetric ? (uint64_t)q_count * (q_count + 1) / 2 : (uint64_t)q_count * c_count;
    profile->kernel_milliseconds = kernel_ms;
    profile->cells = plan.cells;
    profile->pairs = pairs;
    /* Canonical pair-streaming bytes: len(q) + len(c) + two 4-byte offsets + one 8-byte result per pair. */
    profile->algorithmic_bytes = symmetric ? 0 : (uint64_t)c_count * query_bytes + (uint64_t)q_count * candidate_bytes;
    if (symmetric) {
        uint64_t prefix = 0, bytes = 0;
        for (uint32_t i = 0; i < q_count; ++i) prefix += q_lengths[i], bytes += (uint64_t)q_lengths[i] * (i + 1) + prefix;
        profile->algorithmic_bytes = bytes;
    }
    profile->algorithmic_bytes += pairs * 16;
    profile->unique_bytes = query_bytes + (symmetric ? 0 : candidate_bytes) +
                            ((uint64_t)q_count + 1 + (symmetric ? 0 : c_count + 1)) * 4 + (uint64_t)q_count * c_count * 8;
    profile->launches = launches;
    profile->tier = (uint32_t)tier;
    profile->transposed = (uint32_t)transposed;
    profile->cell_bits = cell_bits;
    profile->longest_query = q_longest, profile->longest_candidate = c_longest;
    profile->host_milliseconds = now_milliseconds() - call_started;
    SZS_PHASE(5);
    if (trace)
        fprintf(stderr, "szs call: %.1f us = offsets %.1f + plan %.1f + upload %.1f + launch %.1f + wait %.1f + wrap %.1f | kernel %.1f us\n",
                profile->host_milliseconds * 1e3, phases[0] * 1e3, phases[1] * 1e3, phases[2] * 1e3, phases[3] * 1e3, phases[4] * 1e3,
                phases[5] * 1e3, kernel_ms * 1e3);
#undef SZS_PHASE
    return szs_report(sz_success_k, error_message, NULL);
}
