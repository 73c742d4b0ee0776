/*
 *  fingerprints_probe.c - torch-free parity probe of `szs_fingerprints_*`: seeded random texts through the C-ABI, every
 *  (hash, count) checked against the CPU oracle (oracle/sz_oracle_fingerprints.c).  Test infrastructure.
 *
 *      fingerprints_probe DIMENSIONS TEXTS LEN_LO LEN_HI [REPEATS] [WIDTH ...]       (no widths = the reference's defaults)
 */
#define _POSIX_C_SOURCE 200809L
#define __HIP_PLATFORM_AMD__ 1
#include <hip/hip_runtime_api.h>
#include <signal.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include <unistd.h>

#include "../../include/stringzillas/stringzillas.h"
#include "../../oracle/sz_oracle.h"

static uint64_t rng_state = 0x9E3779B97F4A7C15ull;
static uint64_t rng(void) {
    rng_state ^= rng_state << 13, rng_state ^= rng_state >> 7, rng_state ^= rng_state << 17;
    return rng_state;
}
static double now_ms(void) {
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return ts.tv_sec * 1e3 + ts.tv_nsec * 1e-6;
}
static void on_alarm(int sig) {
    (void)sig;
    fprintf(stderr, "\nHUNG\n");
    _exit(3);
}

int main(int argc, char **argv) {
    if (argc < 5) return fprintf(stderr, "usage: %s DIMENSIONS TEXTS LEN_LO LEN_HI [REPEATS] [WIDTH ...]\n", argv[0]), 2;
    size_t const dimensions = strtoul(argv[1], 0, 10), count = strtoul(argv[2], 0, 10);
    size_t const lo = strtoul(argv[3], 0, 10), hi = strtoul(argv[4], 0, 10);
    int const repeats = argc > 5 ? atoi(argv[5]) : 2;
    size_t widths[16], widths_count = 0;
    for (int i = 6; i < argc && widths_count < 16; ++i) widths[widths_count++] = strtoul(argv[i], 0, 10);
    signal(SIGALRM, on_alarm);
    alarm(getenv("PROBE_ALARM") ? (unsigned)atoi(getenv("PROBE_ALARM")) : 60);

    uint64_t *offsets64 = calloc(count + 1, 8);
    uint32_t *offsets32 = calloc(count + 1, 4);
    size_t total = 0;
    for (size_t i = 0; i < count; ++i) total += lo + rng() % (hi - lo + 1), offsets64[i + 1] = total, offsets32[i + 1] = (uint32_t)total;
    char *data = malloc(total + 1);
    int const binary = getenv("PROBE_BINARY") != NULL;
    for (size_t i = 0; i < total; ++i) data[i] = binary ? (char)(rng() & 0xFF) : "ACGT"[rng() % 4];
    char *device_data;
    uint32_t *device_offsets;
    hipMalloc((void **)&device_data, total + 1), hipMalloc((void **)&device_offsets, (count + 1) * 4);
    hipMemcpy(device_data, data, total, hipMemcpyHostToDevice), hipMemcpy(device_offsets, offsets32, (count + 1) * 4, hipMemcpyHostToDevice);

    size_t const cells = count * dimensions;
    uint32_t *want_hashes = malloc(cells * 4), *want_counts = malloc(cells * 4);
    int const no_oracle = getenv("PROBE_NO_ORACLE") != NULL;
    if (!no_oracle)
        szo_fingerprints_cross(data, offsets64, count, dimensions, 256, widths_count ? widths : NULL, widths_count, 42, want_hashes, want_counts);

    char const *error = NULL;
    szs_device_scope_t scope = NULL;
    sz_status_t status = szs_device_scope_init_gpu_device(0, &scope, &error);
    if (status) return fprintf(stderr, "scope: %d %s\n", status, error ? error : ""), 1;
    sz_capability_t caps;
    szs_device_scope_get_capabilities(scope, &caps, &error);
    szs_fingerprints_t engine = NULL;
    status = szs_fingerprints_init(dimensions, 256, widths_count ? widths : NULL, widths_count, 42, NULL, caps, &engine, &error);
    if (status) return fprintf(stderr, "init: %d %s\n", status, error ? error : ""), 1;

    /* device outputs with a padded stride, and plain host outputs (staged) */
    size_t const padded_stride = dimensions * 4 + 64;
    char *device_hashes, *device_counts;
    hipMalloc((void **)&device_hashes, count * padded_stride), hipMalloc((void **)&device_counts, count * padded_stride);
    uint32_t *got_hashes = malloc(count * padded_stride), *got_counts = malloc(count * padded_stride);
    uint32_t *host_hashes = malloc(cells * 4), *host_counts = malloc(cells * 4);
    sz_sequence_u32tape_t tape = {device_data, device_offsets, count};
    int failures = 0;
    for (int run = 0; run < repeats; ++run) {
        hipMemset(device_hashes, 0xEE, count * padded_stride), hipMemset(device_counts, 0xEE, count * padded_stride);
        double const started = now_ms();
        status = szs_fingerprints_u32tape(engine, scope, &tape, (sz_u32_t *)device_hashes, padded_stride, (sz_u32_t *)device_counts,
                                          padded_stride, &error);
        double const elapsed = now_ms() - started;
        if (status) { printf("run %d: status %d %s\n", run, status, error ? error : ""); ++failures; continue; }
        hipMemcpy(got_hashes, device_hashes, count * padded_stride, hipMemcpyDeviceToHost);
        hipMemcpy(got_counts, device_counts, count * padded_stride, hipMemcpyDeviceToHost);
        size_t bad = 0, padding_touched = 0;
        for (size_t t = 0; t < count && !no_oracle; ++t) {
            uint32_t const *row_h = (uint32_t const *)((char const *)got_hashes + t * padded_stride);
            uint32_t const *row_c = (uint32_t const *)((char const *)got_counts + t * padded_stride);
            for (size_t d = 0; d < dimensions; ++d) {
                int const wrong = row_h[d] != want_hashes[t * dimensions + d] || row_c[d] != want_counts[t * dimensions + d];
                if (wrong && bad < 4)
                    printf(" [text %zu (len %llu) dim %zu: got %u x%u want %u x%u]", t, (unsigned long long)(offsets64[t + 1] - offsets64[t]), d,
                           row_h[d], row_c[d], want_hashes[t * dimensions + d], want_counts[t * dimensions + d]);
                bad += wrong;
            }
            for (size_t d = dimensions; d < dimensions + 16; ++d) padding_touched += row_h[d] != 0xEEEEEEEEu || row_c[d] != 0xEEEEEEEEu;
        }
        double const byte_dims = (double)total * dimensions;
        printf("fingerprints dims %zu texts %zu len[%zu,%zu] run %d: %.3f ms wall, %.1f G byte-dimensions/s, %.2f GB/s of text, %zu bad, %zu padding words touched\n",
               dimensions, count, lo, hi, run, elapsed, byte_dims / elapsed / 1e6, total / elapsed / 1e6, bad, padding_touched);
        failures += bad != 0 || padding_touched != 0;
    }
    if (!no_oracle) { /* the same through plain host outputs */
        status = szs_fingerprints_u32tape(engine, scope, &tape, host_hashes, dimensions * 4, host_counts, dimensions * 4, &error);
        int const same = !status && !memcmp(host_hashes, want_hashes, cells * 4) && !memcmp(host_counts, want_counts, cells * 4);
        printf("host outputs (staged): status %d %s\n", status, same ? "equal" : "DIFFERENT");
        failures += !same;
    }
    szs_fingerprints_free(engine);
    fflush(stdout);
    return failures ? 1 : 0;
}
