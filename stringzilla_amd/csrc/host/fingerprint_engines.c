/*
 *  fingerprint_engines.c - rolling MinHash / Count-Min fingerprint engines: parameter seeding, input normalisation, the segment
 *  plan and the launch.  ROCm counterpart of c/stringzillas/fingerprints.cuh (init, the three call flavours, free) and of
 *  the seeding in include/stringzillas/fingerprints/serial.hpp:495-534.  Nothing here hashes a byte on the CPU.
 *
 *  Per-dimension parameters follow the reference to the bit: multiplier = 256 + splitmix64(seed + dim) % 384, modulo =
 *  4503599626977 - splitmix64(splitmix64(seed + dim)) % 2^20, and the width of dimension `dim` is
 *  `widths[(dim / 64) % count]` when `dimensions` is a whole multiple of 64 x widths (the reference's sliced engines),
 *  else `widths[dim % count]` (its per-dimension fallback) - fingerprints.cuh:49-62,128-176.  What the kernel needs on
 *  top is derived here: the complement (-multiplier^width) mod modulo of the fused rolling update and a reciprocal of the
 *  modulo rounded DOWN (hip/fingerprints.hip).
 */
#include "szs_internal.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

#define SZS_FINGERPRINTS_MAGIC 0x535A5346u

struct szs_fingerprints_s {
    uint32_t magic;
    uint32_t dimensions;
    uint32_t *widths;      /* [dimensions] */
    double *parameters;    /* [4][dimensions]: multipliers, modulos, reciprocals, complements */
    uint32_t widest;

    int device;            /* scratch follows the device of the last call */
    int parameters_device; /* device the parameters were uploaded to, or -1 */
    szs_buffer_t device_parameters; /* [4][dimensions] doubles, then [dimensions] u32 widths */
    szs_buffer_t host_scratch;      /* addresses, lengths */
    szs_buffer_t pinned_staging;    /* refs, prefixes, segment owners, merge list; offsets downloads */
    szs_buffer_t device_tables;     /* the same tables on the device */
    szs_buffer_t device_partials;   /* (double, u32) per (segment of a multi-segment text, dimension) */
    szs_buffer_t device_outputs;    /* dense staging when the caller's outputs are not device-resident */
};

static uint64_t splitmix64(uint64_t state) { /* serial.hpp:44-50; https://prng.di.unimi.it/splitmix64.c */
    state += 0x9E3779B97F4A7C15ull;
    uint64_t z = state;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

sz_status_t szs_fingerprints_create(sz_size_t dimensions, sz_size_t alphabet_size, sz_size_t const *window_widths,
                                    sz_size_t window_widths_count, sz_u64_t seed, sz_capability_t capabilities,
                                    szs_fingerprints_t *out, char const **error_message) {
    (void)alphabet_size; /* the reference's f64 hasher derives nothing from it (serial.hpp:524-533) */
    if (!out) return szs_report(sz_status_unknown_k, error_message, "Engine must not be null");
    if (*out) return szs_report(sz_status_unknown_k, error_message, "Engine must be uninitialized");
    /* Strict by default; with the `cpu_requests` knob set to "gpu" a mask without the GPU bit is served by the only engines
     * this build has (host/tuning.c) - the results are the same numbers, computed on the GPU. */
    if ((capabilities & sz_cap_cuda_k) == 0 && szs_tuning_get(szs_knob_cpu_requests_k) != 1)
        return szs_report(sz_missing_gpu_k, error_message,
                          "The ROCm build ships GPU engines only: request sz_cap_cuda_k (e.g. from a GPU device scope)");
    if ((szs_capabilities() & sz_cap_cuda_k) == 0) return szs_report(sz_missing_gpu_k, error_message, NULL);
    if (!dimensions || dimensions > 0x00FFFFFFu) return szs_report(sz_unexpected_dimensions_k, error_message, NULL);

    static sz_size_t const default_widths[] = {3, 4, 5, 7, 9, 11, 15, 31}; /* fingerprints.cuh:44 */
    if (!window_widths || !window_widths_count) window_widths = default_widths, window_widths_count = 8;
    for (sz_size_t i = 0; i < window_widths_count; ++i)
        if (window_widths[i] < 2 || window_widths[i] > SZS_FINGERPRINT_WIDEST) /* the reference asserts width > 1 */
            return szs_report(sz_unexpected_dimensions_k, error_message, "Window widths must be within [2, 65536]");

    szs_fingerprints_s *engine = (szs_fingerprints_s *)calloc(1, sizeof(szs_fingerprints_s));
    if (engine) engine->widths = (uint32_t *)malloc(dimensions * sizeof(uint32_t));
    if (engine) engine->parameters = (double *)malloc(dimensions * 4 * sizeof(double));
    if (!engine || !engine->widths || !engine->parameters) {
        if (engine) free(engine->widths), free(engine->parameters);
        free(engine);
        return szs_report(sz_bad_alloc_k, error_message, NULL);
    }
    engine->magic = SZS_FINGERPRINTS_MAGIC, engine->dimensions = (uint32_t)dimensions;
    engine->device = -1, engine->parameters_device = -1;

    sz_size_t const per_width_min = dimensions / window_widths_count;
    sz_size_t const per_width_max = (dimensions + window_widths_count - 1) / window_widths_count;
    int const sliced = per_width_min == per_width_max && per_width_min % 64 == 0;
    double *const multipliers = engine->parameters, *const modulos = multipliers + dimensions;
    double *const reciprocals = modulos + dimensions, *const complements = reciprocals + dimensions;
    for (sz_size_t dim = 0; dim < dimensions; ++dim) {
        sz_size_t const width = window_widths[(sliced ? dim / 64 : dim) % window_widths_count];
        double const multiplier = (double)(256ull + splitmix64(seed + dim) % 384ull);
        double const modulo = (double)(4503599626977ull - splitmix64(splitmix64(seed + dim)) % (1ull << 20));
        double power = 1.0; /* multiplier^width mod modulo; every product stays below 2^52, so fmod is exact */
        for (sz_size_t i = 0; i < width; ++i) power = fmod(power * multiplier, modulo);
        engine->widths[dim] = (uint32_t)width;
        multipliers[dim] = multiplier, modulos[dim] = modulo;
        /* strictly below 1 / modulo even after the product with x < 2^52 is rounded: the Barrett quotient never overshoots */
        reciprocals[dim] = (1.0 / modulo) * (1.0 - 0x1p-50);
        complements[dim] = power == 0.0 ? 0.0 : modulo - power;
        if (width > engine->widest) engine->widest = (uint32_t)width;
    }
    *out = engine;
    return szs_report(sz_success_k, error_message, NULL);
}

static void release_device_state(szs_fingerprints_s *engine) {
    szs_buffer_release(&engine->device_parameters);
    szs_buffer_release(&engine->pinned_staging);
    szs_buffer_release(&engine->device_tables);
    szs_buffer_release(&engine->device_partials);
    szs_buffer_release(&engine->device_outputs);
    engine->parameters_device = -1;
}

void szs_fingerprints_destroy(szs_fingerprints_s *engine) {
    if (!engine || engine->magic != SZS_FINGERPRINTS_MAGIC) return;
    if (engine->device >= 0) {
        int previous = 0;
        (void)hipGetDevice(&previous);
        (void)hipSetDevice(engine->device);
        release_device_state(engine);
        (void)hipSetDevice(previous);
    }
    szs_buffer_release(&engine->host_scratch);
    free(engine->widths), free(engine->parameters);
    engine->magic = 0;
    free(engine);
}

sz_status_t szs_fingerprints_call(szs_fingerprints_s *engine, szs_scope_s *scope, szs_input_t const *texts,
                                  sz_u32_t *min_hashes, sz_size_t min_hashes_stride, sz_u32_t *min_counts,
                                  sz_size_t min_counts_stride, char const **error_message) {
    if (!engine || engine->magic != SZS_FINGERPRINTS_MAGIC)
        return szs_report(sz_status_unknown_k, error_message, "Engine must be initialized");
    if (!texts) return szs_report(sz_status_unknown_k, error_message, "Input texts cannot be null");
    int device = 0;
    hipStream_t stream = NULL;
    sz_status_t status = szs_scope_bind_gpu(scope, &device, &stream, error_message);
    if (status != sz_success_k) return status;

    size_t const count = texts->count;
    uint32_t const dimensions = engine->dimensions;
    size_t const row_bytes = (size_t)dimensions * sizeof(uint32_t);
    if (!count) return szs_report(sz_success_k, error_message, NULL);
    if (count > 0x7FFFFFFFull) return szs_report(sz_overflow_risk_k, error_message, NULL);
    if (!min_hashes || !min_counts) return szs_report(sz_status_unknown_k, error_message, "Outputs cannot be null");
    if (min_hashes_stride < row_bytes || min_counts_stride < row_bytes || min_hashes_stride % 4 || min_counts_stride % 4)
        return szs_report(sz_unexpected_dimensions_k, error_message, NULL);

    if (engine->device != device) { /* scratch follows the device of the call */
        if (engine->device >= 0) {
            (void)hipSetDevice(engine->device);
            release_device_state(engine);
            (void)hipSetDevice(device);
        }
        engine->device = device;
    }
    hipError_t error = hipSuccess;
    if (engine->parameters_device != device) { /* [4][dimensions] doubles followed by the widths, once per device */
        size_t const doubles_bytes = (size_t)dimensions * 4 * sizeof(double);
        status = szs_buffer_reserve(&engine->device_parameters, szs_memory_device_k, device, doubles_bytes + row_bytes, error_message);
        if (status != sz_success_k) return status;
        error = hipMemcpy(engine->device_parameters.pointer, engine->parameters, doubles_bytes, hipMemcpyHostToDevice);
        if (error == hipSuccess)
            error = hipMemcpy((char *)engine->device_parameters.pointer + doubles_bytes, engine->widths, row_bytes, hipMemcpyHostToDevice);
        if (error != hipSuccess) return szs_report_hip(error, error_message);
        engine->parameters_device = device;
    }

    /* Host scratch: [addresses][lengths].  Pinned staging: [refs][segment prefix][partial prefix][merge list][segment
     * owners], preceded here by the landing area of an offsets download. */
    uint32_t const texts_count = (uint32_t)count;
    status = szs_buffer_reserve(&engine->host_scratch, szs_memory_host_k, 0, count * (sizeof(uint64_t) + sizeof(uint32_t)), error_message);
    if (status != sz_success_k) return status;
    uint64_t *const addresses = (uint64_t *)engine->host_scratch.pointer;
    uint32_t *const lengths = (uint32_t *)(addresses + count);

    size_t const offsets_bytes = (count + 1) * sizeof(uint64_t);
    status = szs_buffer_reserve(&engine->pinned_staging, szs_memory_pinned_k, device, offsets_bytes, error_message);
    if (status != sz_success_k) return status;
    void const *host_offsets = NULL;
    int download_pending = 0;
    status = szs_prefetch_offsets(engine->pinned_staging.pointer, stream, texts, 0, &host_offsets, &download_pending, error_message);
    if (status != sz_success_k) return status;
    if (download_pending) {
        error = hipStreamSynchronize(stream);
        if (error != hipSuccess) return szs_report_hip(error, error_message);
    }
    uint64_t total_bytes = 0;
    status = szs_gather_strings(texts, host_offsets, addresses, lengths, &total_bytes, NULL, error_message);
    if (status != sz_success_k) return status;

    /* The segment plan: a text is cut into stretches of SZS_FINGERPRINT_SEGMENT window positions hashed independently. */
    uint64_t total_segments = 0, partial_slots = 0;
    uint32_t merge_count = 0;
    for (uint32_t i = 0; i < texts_count; ++i) {
        uint32_t const segments = lengths[i] ? (lengths[i] + SZS_FINGERPRINT_SEGMENT - 1) / SZS_FINGERPRINT_SEGMENT : 1;
        total_segments += segments;
        if (segments > 1) partial_slots += segments, ++merge_count;
    }
    if (total_segments > 0xFFFFFFF0ull) return szs_report(sz_overflow_risk_k, error_message, NULL);
    int const needs_owners = total_segments != texts_count;

    size_t const refs_at = 0, segment_prefix_at = refs_at + (size_t)texts_count * sizeof(szs_string_ref_t);
    size_t const partial_prefix_at = segment_prefix_at + ((size_t)texts_count + 1) * sizeof(uint32_t);
    size_t const merge_list_at = partial_prefix_at + ((size_t)texts_count + 1) * sizeof(uint32_t);
    size_t const owners_at = merge_list_at + ((size_t)merge_count + 1) * sizeof(uint32_t);
    size_t const tables_bytes = owners_at + (needs_owners ? (size_t)total_segments * sizeof(uint32_t) : 0);
    /* the gathered offsets have been consumed: the staging area may be re-reserved (contents are not preserved) */
    status = szs_buffer_reserve(&engine->pinned_staging, szs_memory_pinned_k, device, tables_bytes > offsets_bytes ? tables_bytes : offsets_bytes,
                                error_message);
    if (status != sz_success_k) return status;
    status = szs_buffer_reserve(&engine->device_tables, szs_memory_device_k, device, tables_bytes, error_message);
    if (status != sz_success_k) return status;
    char *const host_tables = (char *)engine->pinned_staging.pointer, *const device_tables = (char *)engine->device_tables.pointer;
    szs_string_ref_t *const refs = (szs_string_ref_t *)(host_tables + refs_at);
    uint32_t *const segment_prefix = (uint32_t *)(host_tables + segment_prefix_at);
    uint32_t *const partial_prefix = (uint32_t *)(host_tables + partial_prefix_at);
    uint32_t *const merge_list = (uint32_t *)(host_tables + merge_list_at);
    uint32_t *const owners = (uint32_t *)(host_tables + owners_at);
    uint32_t segment_cursor = 0, partial_cursor = 0, merge_cursor = 0;
    for (uint32_t i = 0; i < texts_count; ++i) {
        refs[i].address = addresses[i], refs[i].length = lengths[i], refs[i].index = i;
        uint32_t const segments = lengths[i] ? (lengths[i] + SZS_FINGERPRINT_SEGMENT - 1) / SZS_FINGERPRINT_SEGMENT : 1;
        segment_prefix[i] = segment_cursor, partial_prefix[i] = partial_cursor;
        if (needs_owners)
            for (uint32_t s = 0; s < segments; ++s) owners[segment_cursor + s] = i;
        segment_cursor += segments;
        if (segments > 1) partial_cursor += segments, merge_list[merge_cursor++] = i;
    }
    segment_prefix[texts_count] = segment_cursor, partial_prefix[texts_count] = partial_cursor;
    error = hipMemcpyAsync(device_tables, host_tables, tables_bytes, hipMemcpyHostToDevice, stream);
    if (error != hipSuccess) return szs_report_hip(error, error_message);

    if (partial_slots) {
        status = szs_buffer_reserve(&engine->device_partials, szs_memory_device_k, device,
                                    (size_t)partial_slots * dimensions * (sizeof(double) + sizeof(uint32_t)), error_message);
        if (status != sz_success_k) return status;
    }
    double *const partial_minimums = (double *)engine->device_partials.pointer;
    uint32_t *const partial_counts = partial_slots ? (uint32_t *)(partial_minimums + (size_t)partial_slots * dimensions) : NULL;

    /* Outputs in device memory are written in place; anything else is staged densely and copied out row by row. */
    int const direct = szs_classify_pointer(min_hashes).device_resident && szs_classify_pointer(min_counts).device_resident;
    uint32_t *device_hashes = min_hashes, *device_counts = min_counts;
    size_t hashes_stride = min_hashes_stride, counts_stride = min_counts_stride;
    if (!direct) {
        status = szs_buffer_reserve(&engine->device_outputs, szs_memory_device_k, device, 2 * count * row_bytes, error_message);
        if (status != sz_success_k) return status;
        device_hashes = (uint32_t *)engine->device_outputs.pointer, device_counts = device_hashes + count * dimensions;
        hashes_stride = counts_stride = row_bytes;
    }

    double const *const device_doubles = (double const *)engine->device_parameters.pointer;
    int const launch_error = szs_hip_fingerprints(
        (szs_string_ref_t const *)(device_tables + refs_at), texts_count,
        needs_owners ? (uint32_t const *)(device_tables + owners_at) : NULL, (uint32_t const *)(device_tables + segment_prefix_at),
        (uint32_t const *)(device_tables + partial_prefix_at), (uint32_t)total_segments,
        (uint32_t const *)(device_tables + merge_list_at), merge_count, dimensions,
        (uint32_t const *)(device_doubles + (size_t)dimensions * 4), device_doubles, device_doubles + dimensions,
        device_doubles + (size_t)dimensions * 2, device_doubles + (size_t)dimensions * 3, partial_minimums, partial_counts,
        device_hashes, hashes_stride, device_counts, counts_stride, engine->widest, stream);
    if (launch_error) return szs_report_hip((hipError_t)launch_error, error_message);

    if (!direct) {
        error = hipMemcpy2DAsync(min_hashes, min_hashes_stride, device_hashes, row_bytes, row_bytes, count, hipMemcpyDefault, stream);
        if (error == hipSuccess)
            error = hipMemcpy2DAsync(min_counts, min_counts_stride, device_counts, row_bytes, row_bytes, count, hipMemcpyDefault, stream);
    }
    if (error == hipSuccess) error = hipStreamSynchronize(stream); /* the call is synchronous, like the reference's */
    if (error != hipSuccess) return szs_report_hip(error, error_message);
    return szs_report(sz_success_k, error_message, NULL);
}
